"""LayerNorm forward at d = 768: the row-pair kernel with the next pair prefetched (round 6, ln_fwd_pair768_kernel) against the
one-row-at-a-time register kernel (EVLM_LN_FWD_NO_PAIR=1) on the step's shapes, streaming from HBM (16 distinct inputs in
rotation: 0.3 - 0.9 GB per sweep, beyond the 256 MB last-level cache).  python tools/ln_fwd_pair_bench.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import torch
    from efficientvlm_amd import ops
    res = {}
    for rows in (12608, 36928, 28832, 7680):
        xs = [torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16) for _ in range(16)]
        w = torch.randn(768, device="cuda"); b = torch.randn(768, device="cuda")
        with torch.no_grad():
            for x in xs[:3]:
                ops.layer_norm(x, w, b, 1e-5)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for x in xs:
                    ops.layer_norm(x, w, b, 1e-5)
            g.replay(); torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                g.replay()
            e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) / (20 * 16) * 1e3
        res[rows] = {"us": round(us, 2), "TB/s": round(rows * 768 * 2 * 2 / 1e6 / us, 2)}
    print(json.dumps(res))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for flag, name in (("1", "one row per trip (ln_fwd_reg_kernel)"), ("", "row pairs, prefetched (ln_fwd_pair768_kernel)")):
            env = dict(os.environ)
            env.pop("EVLM_LN_FWD_NO_PAIR", None)
            if flag:
                env["EVLM_LN_FWD_NO_PAIR"] = flag
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
            print(name, out[-1] if out else "FAILED")
