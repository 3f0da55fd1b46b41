#!/usr/bin/env python3
"""CU-masked streams for the teacher / student co-scheduling of the GD step.

hipExtStreamCreateWithCUMask gives a stream whose kernels may only run on the CUs of its mask.  A hipGraph's internal
branches do not inherit stream attributes (round 4: priorities), so the joint graph cannot be masked; the masked form is
TWO graphs - the teacher forward of the new batch on a stream restricted to T CUs, the student step of the waiting batch on a
stream restricted to the other 256 - T - launched side by side.  Reported, same process and box:

  1. what a mask does to one chip-filling GEMM launch (does the mask bind, and how the bits map to CUs);
  2. ms/step of: the joint graph (shipping), two graphs unmasked, two graphs with T = 32 .. 128 teacher CUs
     (student's text-stream fork off inside the masked graphs: an internal branch would escape the mask).

    python tools/cu_mask_probe.py [--steps 20]"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from efficientvlm_amd import ops
from efficientvlm_amd.trainer import no_gc_during_capture
from efficientvlm_amd.workload import GEOMS, make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--mode", default="map", choices=["map", "joint", "two", "masked"],
                help="map: which CUs a mask selects + one GEMM under it; joint / two / masked: ms per GD step of that form (ONE form "
                     "per process: streams share a few hardware queues, a masked stream created earlier colours later ones)")
ap.add_argument("--teacher-cus", type=int, default=64)
ap.add_argument("--layout", default="block", choices=["block", "striped", "xcd"])
ap.add_argument("--serial-text", action="store_true", help="student text pass / distillation terms in sequence (no internal forks)")
args = ap.parse_args()
if args.serial_text:
    os.environ["EVLM_NO_TEXT_STREAM"] = "1"; os.environ["EVLM_NO_KD_STREAM"] = "1"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """stream restricted to the CUs whose bit is set (list of CU indices)"""
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


def time_on(stream, fn, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cu_histogram(stream):
    """distinct (xcc, se, sh, cu) ids that 8 192 spinning workgroups launched on `stream` ran on"""
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libcuprobe.so")
    if not os.path.exists(so):
        import subprocess
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                        os.path.join(os.path.dirname(os.path.abspath(__file__)), "cu_probe.hip")], check=True)
    lib = ctypes.CDLL(so)
    n = 8192
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = lib.cu_probe(ctypes.c_void_p(out.data_ptr()), n, 20000, ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().view(n, 2)
    hw, xcc = o[:, 0].long() & 0xFFFFFFFF, o[:, 1].long() & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = [sum(1 for i in ids if i[0] == x) for x in range(8)]
    return len(ids), per_xcc


def layout_bits(tc, layout):
    if layout == "block":
        tb = list(range(tc))
    elif layout == "striped":
        stride = NCU // tc
        tb = [i for i in range(NCU) if i % stride == 0][:tc]
    else:                                       # "xcd": the same tc / 8 CUs of every XCD, if bit i is CU i // 8 of XCD i % 8
        tb = list(range(tc))
    return tb, [i for i in range(NCU) if i not in set(tb)]


if args.mode == "map":
    # ---- 1. which CUs does a mask select, and does it bind? ----------------------------------------------------------
    x = (torch.randn(12608, 768, device=dev) * 0.1).bfloat16()
    w = torch.nn.Parameter(torch.randn(3072, 768, device=dev) * 0.03, requires_grad=False)
    gemm = lambda: ops.linear(x, w, None)
    with torch.no_grad():
        plain = torch.cuda.Stream()
        base = time_on(plain, gemm)
        print(f"{NCU} CUs; unmasked stream: ViT FC1 12608 x 3072 x 768 {base:.1f} us; workgroups ran on {cu_histogram(plain)}", flush=True)
        for name, bits in (("first 128 bits", range(128)), ("even bits", range(0, NCU, 2)), ("first 64 bits", range(64)),
                           ("bits = 0 mod 4", range(0, NCU, 4)), ("first 32 bits", range(32)), ("bits 32..63", range(32, 64)),
                           ("first 8 bits", range(8)), ("bits = 0 mod 8", range(0, NCU, 8)), ("bits 0..7 + 128..135", list(range(8)) + list(range(128, 136)))):
            st = masked_stream(list(bits))
            t = time_on(st, gemm)
            n, per = cu_histogram(st)
            print(f"   mask {name:22s} ({len(list(bits)):3d} bits): GEMM {t:7.1f} us ({t / base:.2f} x); ran on {n:3d} CUs, per XCD {per}", flush=True)
    sys.exit(0)

# ---- 2. the GD step ----------------------------------------------------------------------------------------------------
geom = GEOMS["full"]
batches = [{k: v.to(dev) for k, v in make_batch(geom, 64, seed=42 + 1000 * i).items()} for i in range(4)]


def joint():
    s, t = bench.build(geom, dev, 1234)
    tr = bench.make_trainer(s, t, torch.bfloat16, True, True)
    it = 0
    for _ in range(7):
        tr.step(batches[it % 4]); it += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(batches[it % 4]); it += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    tr.close()
    return ms


def two_graphs(stream_t, stream_s, tag):
    s, t = bench.build(geom, dev, 1234)
    tr = bench.make_trainer(s, t, torch.bfloat16, True, True)
    for i in range(3):                            # creates the batch kind's static buffers, warms allocator and caches
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    (pipe,) = tr._pipes.values()
    cur = torch.cuda.current_stream()
    gT, gS, outs = [], [], []
    ops.CACHE.invalidate(); ops.reserve_tables()
    for k in (0, 1):
        g = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(g, stream=stream_t, capture_error_mode="thread_local"):
            tr._teacher_eager(pipe, k)
        gT.append(g)
    for k in (0, 1):
        tr.opt.set_schedule(1.0)
        g = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(g, stream=stream_s, capture_error_mode="thread_local"):
            outs.append(tr._student_eager(pipe, k))
        gS.append(g)
    ops.flush_table_uploads()
    torch.cuda.synchronize()

    def step(batch, p):
        for name, v in batch.items():
            pipe["B"][p][name].copy_(v, non_blocking=True)
        tr.opt.set_schedule(1.0)
        stream_t.wait_stream(cur); stream_s.wait_stream(cur)
        with torch.cuda.stream(stream_t):
            gT[p].replay()
        with torch.cuda.stream(stream_s):
            gS[1 - p].replay()
        cur.wait_stream(stream_t); cur.wait_stream(stream_s)
        tr.opt._scheduled = False

    p, it = 0, 0
    for _ in range(6):
        p = 1 - p; step(batches[it % 4], p); it += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        p = 1 - p; step(batches[it % 4], p); it += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    loss = [round(float(v), 3) for v in outs[1 - p].tolist()]
    print(f"{tag:64s} {ms:7.2f} ms/step   losses {loss}", flush=True)
    tr.close()
    return ms


if args.mode == "joint":
    print(f"{'joint hipGraph' + (' (text pass in sequence)' if args.serial_text else ' (shipping)'):64s} {joint():7.2f} ms/step", flush=True)
elif args.mode == "two":
    two_graphs(torch.cuda.Stream(), torch.cuda.Stream(), "two graphs, unmasked streams" + (", text pass in sequence" if args.serial_text else ""))
else:
    tb, sb = layout_bits(args.teacher_cus, args.layout)
    st_t, st_s = masked_stream(tb), masked_stream(sb)
    nt, ns = cu_histogram(st_t)[0], cu_histogram(st_s)[0]
    two_graphs(st_t, st_s, f"two graphs, teacher mask {len(tb)} bits ({args.layout}) -> {nt} CUs, student {len(sb)} bits -> {ns} CUs")
