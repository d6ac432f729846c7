"""bench.py started WITHOUT a launcher: `--gpus N` must bring up its own N ranks (or fail loudly), never print n_gpus 1
for a multi-GPU request.  The ranks share the box's one GPU over gloo here (GCMF_BENCH_SHARE_GPU=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_8_without_devices_fails_loudly():
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "only" in (r.stderr + r.stdout) and "HIP device" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("cfg,extra", [(3, []), (4, ["--scaling", "weak"]), (5, ["--nlev", "5"]), (3, ["--exchange", "p2p"]),
                                       (4, ["--exchange", "p2p"])])
def test_self_launch_two_ranks_sharing_the_gpu(cfg, extra):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", str(cfg), "--ny", "256", "--nx", "256",
              "--no-weak", *extra], {"GCMF_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["scaling"] == ("weak" if "weak" in extra else "strong")
    assert line["config"]["global_grid"] == [512 if "weak" in extra else 256, 256]
    if cfg != 5:
        assert 0 < line["roofline"]["frac"] < 1      # physical: algorithmic bytes of a launch x launches / time, like the N = 1 line
    if cfg != 5:   # the N > 1 line says what its halo exchanges cost (host + device, per exchange)
        ex = line["exchange"]
        assert ex["kind"] == ("p2p" if "p2p" in extra else "torch") and ex["exchanges_per_application"] >= 1
        assert ex["ms_per_application_without_exchange"] > 0 and ex["us_per_exchange_host_and_device"] is not None


@pytest.mark.parametrize("cfg,exchange", [(3, "torch"), (3, "p2p"), (4, "p2p")])
def test_two_ranks_with_skewed_clocks_run_matched_collectives(cfg, exchange):
    """Round 3's bench let every rank extend its warm-up by its OWN clock: a rank that arrived late ran one collective application
    more than its neighbour (p2p: a chain of time-outs; RCCL: an unmatched recv = a hang).  Rank 1 is held back 30 ms before every
    timed() here; the ranks must still run identical numbers of applications and exchanges, and nothing may time out."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", str(cfg), "--ny", "256", "--nx", "256", "--no-weak",
              "--exchange", exchange], {"GCMF_BENCH_SHARE_GPU": "1", "GCMF_BENCH_SKEW_MS": "30", "GCMF_P2P_TIMEOUT_MS": "5000"})
    assert r.returncode == 0, r.stderr[-2000:]
    ex = _json_line(r.stdout)["exchange"]
    assert ex["kind"] == exchange and ex["matched_across_ranks"] is True
    calls = ex["collective_calls_rank0"]
    assert calls["timed"] == 2 * 3 and calls["warmup"] >= 2 and calls["exchanges"] > 0
    if exchange == "p2p":
        assert calls["p2p_seq"] > 0


def test_hung_multi_rank_run_is_killed_by_the_parent():
    """The un-initialised parent of `bench.py --gpus N` is the watchdog: a run that exceeds GCMF_BENCH_TIMEOUT_S is killed as a
    process group and reported with exit code 124 (never a hang until the driver's own limit)."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "3", "--ny", "256", "--nx", "256", "--no-weak"],
             {"GCMF_BENCH_SHARE_GPU": "1", "GCMF_BENCH_TIMEOUT_S": "0.5"}, timeout=120)
    assert r.returncode == 124 and "killing its process group" in r.stderr


def test_single_gpu_line_carries_parity_and_roofline():
    r = _run(["--steps", "2", "--warmup", "1", "--no-extra", "--cpu-steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["n_steps"] == 63
    par = line["parity"]
    assert par["rel_err"] <= 1e-6 and par["nan_pattern_equal"] is True
    assert par["reference_probes"]["rel_err"] <= 1e-6 and par["reference_probes"]["n_probes"] == 300
    rf = line["roofline"]
    # the contract's roofline is a hardware fraction: algorithmic bytes of one launch / launch time / 8 TB/s
    assert rf["kernel"].startswith("gcmf::k_") and rf["alg_bytes_per_launch"] > 0 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - rf["alg_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * rf["achieved"]
    assert rf["alg_one_pass_per_step_frac"] > rf["frac"] and rf["geometry"]["H"] > 0
    if rf["traffic"] is not None:   # counter bytes are quoted only for the kernel AND launch geometry that were profiled
        assert rf["traffic_source"].startswith("profiles/") and rf["frac"] <= rf["hbm_frac"] < 1
        assert str(rf["geometry"]["H"]) in rf["traffic_source"]
    for k in ("value_min", "value_max", "value_mean"):
        assert line[k] > 0
    assert line["value_min"] <= line["value"] <= line["value_max"]
    assert line["timing"]["blocks"] == 2 and line["timing"]["plans_refolded_between_blocks"] == 1
    assert line["cpu_baseline"]["cores"] == 1
    # round 6: the line ENDS with a compact summary (the driver's record keeps the tail): [G cell-steps/s, frac, traffic / algorithmic, parity]
    assert list(line)[-1] == "summary" and len(json.dumps(line["summary"])) <= 1500
    row = line["summary"]["cfg3"]
    assert abs(row[0] - line["value"] / 1e9) < 0.06 and abs(row[1] - rf["frac"]) < 1e-3 and row[3] <= 1e-6


def test_kinds_line_has_a_roofline_per_kind():
    """`bench.py --kinds` (tools/bench_kinds.py): one record per Laplacian kind that is not a BASELINE config, each with
    the dominant kernel, its launch time, `frac` and parity against the oracle (here on a small grid and two kinds, to keep it short)."""
    r = _run(["--kinds", "--kinds-only", "mom5u,regular_area", "--ny", "240", "--nx", "360"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert [k["kind"] for k in line["kinds"]] == ["mom5u", "regular_area"] and list(line)[-1] == "summary"
    for k in line["kinds"]:
        rf = k["roofline"]
        assert k["value"] > 0 and rf["kernel"].startswith("gcmf::k_") and 0 < rf["frac"] < 1 and rf["launches_of_it_per_application"] >= 1
        assert k["parity"]["rel_err"] <= 1e-12 and k["parity"]["nan_pattern_equal"] is True
