"""GPU-box experiments behind DESIGN 3.2's stream findings, on C2's model.
    stream_kind_test.py           cost of running the factorisation on created streams of different kind / priority
    stream_kind_test.py waits     event waits against the legacy default (null) stream vs created streams (the ~7 ms per
                                  evaluation finding; was stream_kind_test2.py)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gptorch_amd import _native, _ops
dev = torch.device("cuda:0")
m, _, _ = bench.build_model(bench.WORKLOADS["c2"], 0, dev)


def kinds():
    def run(stream, variant):
        _native.debug_begin().gpn_debug_set_potrf_variant(variant)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx, torch.no_grad():
            for _ in range(3): m.log_likelihood()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m.log_likelihood()
            torch.cuda.synchronize()
        _native.debug_end()
        return (time.perf_counter() - t0) / 10 * 1e3
    s_norm = torch.cuda.Stream(device=dev)
    s_hi = torch.cuda.Stream(device=dev, priority=-1)
    for name, st in (("default", None), ("created", s_norm), ("created-hi", s_hi)):
        print("%-11s look-ahead %.2f ms   recursion %.2f ms" % (name, run(st, 0), run(st, 1)), flush=True)


def waits():
    k = m.kernel
    cur = torch.cuda.current_stream(dev)
    st = torch.cuda.Stream(device=dev)
    def one(stream, waits):
        with torch.no_grad():
            if waits: stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                resid = m.Y - m.mean_function(m.X)
                f = _ops.kernel_factor_async(k._kind, m.X, k.variance.transform(), k.length_scales.transform(),
                                             m.likelihood.variance.transform(), R=resid, factor=m._holder.get("factor"))
                m._holder["factor"] = f
                t = f.lml_terms()
            if waits: cur.wait_stream(stream)
            return int(f.info.item()), t
    main2 = torch.cuda.Stream(device=dev)
    for name, stream, waits, mainstream in (("cur", cur, False, None), ("created+waits (null main)", st, True, None), ("created, no waits", st, False, None),
                                            ("created+waits (created main)", st, True, main2)):
        if mainstream is not None:
            torch.cuda.set_stream(mainstream)
            cur = mainstream
        for _ in range(3): one(stream, waits)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): one(stream, waits)
        torch.cuda.synchronize()
        print("%-30s %.2f ms" % (name, (time.perf_counter() - t0) / 10 * 1e3), flush=True)


(waits if len(sys.argv) > 1 and sys.argv[1] == "waits" else kinds)()
