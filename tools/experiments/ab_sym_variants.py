"""Round 6: A/B of pair-symmetric filter variant libraries (tools/experiments/variants/NAME.so) on one box.
Each library runs in its own process (one library per process: the loader binds at import): back-to-back time of the 1080p r = 20
launch (best of 3 x 20), the same launch right behind a 256-spp-sized accumulation (the in-step situation: held clock), and the
output's relative L2 against the first library's output (the variants change the order of the sums, not the spec).
  python tools/experiments/ab_sym_variants.py base gs5 gs6 ...        (GPU box)
  python tools/experiments/ab_sym_variants.py --one NAME OUT.pt       (internal)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "tools", "experiments", "variants")


def one(name, out_path):
    from statmc_amd import build
    os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1")
    build.SO = os.path.join(VAR, name + ".so")
    import torch
    from statmc_amd import api, film, synthetic
    W, H = 1920, 1080
    dev = torch.device("cuda:0")
    api.setup(0)
    scene = synthetic.Scene(W, H, seed=1, device=dev)
    fs = film.FilmStats(W, H, dev)
    smp = scene.samples(32, seed=2, features=("radiance", "normal", "albedo"))
    fs.accumulate(smp)
    fs.prepass()
    api.force_filter_parts(1)
    a, keep = fs.filter_args()

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for _ in range(3):
        api.window_filter(a, 3)
    torch.cuda.synchronize()
    b2b = min(timed(lambda: api.window_filter(a, 3), 20) for _ in range(3))
    out = fs.film_f.clone()
    # behind a long memory-bound kernel: 8 x 32 spp of accumulation (about the 256-spp launch's 3.7 ms), then ONE filter, timed alone
    fs2 = film.FilmStats(W, H, dev)
    ts = []
    for _ in range(12):
        for _ in range(8):
            fs2.accumulate(smp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        api.window_filter(a, 3)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    torch.save(out.cpu(), out_path)
    print("%-12s back to back %.3f ms | behind the accumulation %.3f ms (median of 10; min %.3f) | variant %s"
          % (name, b2b, ts[len(ts) // 2], ts[0], api.last_filter_variant()), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], sys.argv[3])
        sys.exit(0)
    import torch
    names = sys.argv[1:]
    outs = {}
    for rnd in range(2):       # two passes in alternating order: a drifting box shows
        for n in (names if rnd == 0 else names[::-1]):
            p = "/tmp/ab_%s.pt" % n
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", n, p], capture_output=True, text=True, timeout=300)
            sys.stdout.write(r.stdout)
            if r.returncode != 0:
                print(n, "FAILED", r.stderr[-800:], flush=True)
                continue
            outs[n] = torch.load(p)
    base = outs.get(names[0])
    for n in names:
        if n in outs and base is not None:
            d = (outs[n].double() - base.double())
            print("%-12s rel L2 vs %s: %.3e   max abs %.3e   finite %s" % (n, names[0], float((d.pow(2).sum() / base.double().pow(2).sum()).sqrt()), float(d.abs().max()), bool(torch.isfinite(outs[n]).all())))
