"""Does an EXTERNAL event recorded INSIDE a captured hipGraph release a stream that waits for it from OUTSIDE the graph, at
the point of the graph where it was recorded (round 6: the N > 1 step could then keep its whole backward in ONE graph and let
the reducer stream pick the gradient stages up through events instead of cutting the graph at every stage)?
   graph:  sleep 4 ms -> record(ev, hipEventRecordExternal) -> sleep 4 ms          other stream: wait(ev) -> timestamp
expected if it works: the other stream's timestamp ~4 ms after the graph's start; ~8 ms = released only at the graph's end;
~0 ms = the wait saw an already-complete event (no dependency).  torch.cuda.Event(external=True) is refused on ROCm builds, so the
record goes through hipEventRecordWithFlags directly (ctypes on libamdhip64)."""
import ctypes as C, sys, time, torch
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_ms = 10_000_000 / e0.elapsed_time(e1)
S = int(4 * cyc_ms)
ev = C.c_void_p()
assert hip.hipEventCreateWithFlags(C.byref(ev), 2) == 0          # hipEventDisableTiming
cs, other = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cs):
    torch.cuda._sleep(S)
    rc = hip.hipEventRecordWithFlags(ev, C.c_void_p(cs.cuda_stream), 1)       # hipEventRecordExternal
    print("hipEventRecordWithFlags(external) during capture: rc", rc, "last error", hip.hipGetLastError(), flush=True)
    if rc == 0:
        torch.cuda._sleep(S)
if rc != 0:
    # (round 6, ROCm 7.2: rc 1 = hipErrorInvalidValue.  Inserting the node through the graph API instead - hipStreamGetCaptureInfo_v2
    # -> hipGraphAddEventRecordNode -> hipStreamUpdateCaptureDependencies - crashed the process.)
    print("this HIP runtime refuses an external event record inside a stream capture: the graph cannot hand a gradient stage "
          "over through an event - the N > 1 step keeps its cuts")
    sys.exit(0)
for wait_flag in (0, 1):
    for rep in range(3):
        torch.cuda.synchronize()
        t_start, t_other, t_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(cs):
            t_start.record()
            g.replay()
            t_end.record()
        with torch.cuda.stream(other):
            rcw = hip.hipStreamWaitEvent(C.c_void_p(other.cuda_stream), ev, wait_flag)
            t_other.record()
        torch.cuda.synchronize()
        print(f"wait flag {wait_flag} rep {rep} (rc {rcw}): other stream released {t_start.elapsed_time(t_other):.2f} ms after the graph started; graph took {t_start.elapsed_time(t_end):.2f} ms")
