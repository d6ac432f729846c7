"""Host-side logic of the row-block pipeline (gcm_filters_amd/host_blocks.py) that needs no GPU: how many blocks, which rows of
the caller's array a block (with its n_steps ghost rows, periodic in y) is filled from."""
import numpy as np
import pytest

from gcm_filters_amd import host_blocks as hb


def test_choose_blocks(monkeypatch):
    monkeypatch.delenv("GCMF_HOST_BLOCKS", raising=False)
    assert hb.choose_blocks(2400, 56) == 3          # BASELINE config 2: 800 rows per block >= 14 * 56
    assert hb.choose_blocks(2400, 63) == 2          # config 3: ghost zones of 63 rows cost too much for three blocks
    assert hb.choose_blocks(2400, 700) == 0         # ghost rows would outnumber the block's own rows
    assert hb.choose_blocks(200, 5) == 0            # blocks of fewer than 128 rows are not worth a plan each
    monkeypatch.setenv("GCMF_HOST_BLOCKS", "0")
    assert hb.choose_blocks(2400, 56) == 0
    monkeypatch.setenv("GCMF_HOST_BLOCKS", "1")     # one block is the plain path
    assert hb.choose_blocks(2400, 56) == 0
    monkeypatch.setenv("GCMF_HOST_BLOCKS", "5")
    assert hb.choose_blocks(2400, 56) == 5
    assert hb.choose_blocks(2400, 400) == 3         # shrinks until a block holds 2 * n_steps rows
    assert hb.choose_blocks(300, 10) == 2


@pytest.mark.parametrize("ny,nblocks,ghost", [(2400, 3, 56), (301, 2, 17), (600, 4, 40)])
def test_block_rows_cover_the_periodic_grid(ny, nblocks, ghost):
    field = np.arange(ny)
    base, rem = divmod(ny, nblocks)
    b = 0
    for k in range(nblocks):
        e = b + base + (1 if k < rem else 0)
        rows = (e - b) + 2 * ghost
        got = np.empty(rows, dtype=field.dtype)
        runs = hb.block_runs(b - ghost, rows, ny)
        assert 1 <= len(runs) <= 2                           # a block wraps around the seam at most once
        for r, gj, n in runs:
            got[r: r + n] = field[gj: gj + n]
        assert np.array_equal(got, (np.arange(b - ghost, e + ghost)) % ny)
        b = e
    assert b == ny
