"""Do an upload and a download overlap on this box?  (the host path of config 3 moves 69 MB each way per field: 1.24 + 1.25 ms in sequence)"""
import threading, time, numpy as np, torch
n = 2400 * 3600
d_in = torch.empty(n, dtype=torch.float64, device="cuda"); d_out = torch.rand(n, dtype=torch.float64, device="cuda")
p_in = torch.rand(n, dtype=torch.float64).pin_memory(); p_out = torch.empty(n, dtype=torch.float64).pin_memory()
g_in = torch.rand(n, dtype=torch.float64); g_out = torch.empty(n, dtype=torch.float64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return 1e3 * min(t)
def up(h):
    with torch.cuda.stream(s1): d_in.copy_(h, non_blocking=True)
def down(h):
    with torch.cuda.stream(s2): h.copy_(d_out, non_blocking=True)
print(f"pinned:   H2D {timed(lambda: up(p_in)):.2f} ms   D2H {timed(lambda: down(p_out)):.2f} ms   both streams at once {timed(lambda: (up(p_in), down(p_out))):.2f} ms")
def both_threads(hi, ho):
    th = threading.Thread(target=lambda: (down(ho), s2.synchronize()))
    th.start(); up(hi); s1.synchronize(); th.join()
print(f"pageable: H2D {timed(lambda: up(g_in)):.2f} ms   D2H {timed(lambda: down(g_out)):.2f} ms   two host threads at once {timed(lambda: both_threads(g_in, g_out)):.2f} ms")
print(f"pinned, two host threads at once {timed(lambda: both_threads(p_in, p_out)):.2f} ms")
# chunks of 8 MB alternately (what a pipeline does)
ch = 1 << 20
def chunks():
    for o in range(0, n, ch):
        with torch.cuda.stream(s1): d_in[o:o + ch].copy_(p_in[o:o + ch], non_blocking=True)
        with torch.cuda.stream(s2): p_out[o:o + ch].copy_(d_out[o:o + ch], non_blocking=True)
print(f"pinned, 8-MB chunks alternating on two streams {timed(chunks):.2f} ms")
# pageable memory, two host threads, chunked: does a blocking upload overlap with a blocking download?
for mb in (2, 8, 32):
    chn = mb << 17
    def up_chunks(h):
        for o in range(0, n, chn):
            with torch.cuda.stream(s1): d_in[o:o + chn].copy_(h[o:o + chn], non_blocking=True)
        s1.synchronize()
    def down_chunks(h):
        for o in range(0, n, chn):
            with torch.cuda.stream(s2): h[o:o + chn].copy_(d_out[o:o + chn], non_blocking=True)
        s2.synchronize()
    def both(hi, ho):
        th = threading.Thread(target=down_chunks, args=(ho,)); th.start(); up_chunks(hi); th.join()
    print(f"{mb:2d}-MB chunks, two host threads: pageable {timed(lambda: both(g_in, g_out)):.2f} ms   pinned {timed(lambda: both(p_in, p_out)):.2f} ms   "
          f"(pageable up alone {timed(lambda: up_chunks(g_in)):.2f}, down alone {timed(lambda: down_chunks(g_out)):.2f})")
