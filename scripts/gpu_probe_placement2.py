"""Scratch probe (not a test): iteration time of the solver against the physical placement of its workspace.
Round A: workspace re-allocated per trial (previous one freed).  Round B: four workspaces held at once, each timed twice;
the best is kept, the others are freed, and it is timed again (is a placement's speed a stable property?)."""
import importlib
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
Z, Y, X = 64, 2048, 2048
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
n_em = Z * Y * X // 3400  # the bench's kind of image (a flat one stops on its KL plateau after three iterations)
vol = torch.zeros((Z, Y, X), device=dev)
idx = torch.randint(0, Z * Y * X, (n_em,), device=dev, generator=g)
vol.view(-1).index_put_((idx,), (torch.rand(n_em, device=dev, generator=g) * 600 + 300) / 0.0138, accumulate=True)
kernels.gaussian_lowpass(vol, (1.8, 1.6, 1.6), True, out=vol)
img = torch.poisson(vol + 100.0, generator=g).clamp_(0, 65535).to(torch.uint16)
del vol
torch.cuda.empty_cache()
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (64, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
psf = (psf / psf.sum()).float()
out = torch.empty((Z, Y, X), dtype=torch.float32, device=dev)


def timed(n=12):
    _, info = rl.rlgc_device(img, psf, max_iterations=n, out=out, rng_seed=1, profile="loop")
    assert info.iterations == n, info
    return info.loop_ms / (n + 0.3)  # the loop opens with the first forward model


def fresh_plan():
    """A new plan + workspace; the old plan object is destroyed, its workspace tensor lives while the caller holds it."""
    old = list(rl._plan_cache.values())
    rl._plan_cache.clear()
    rl.rlgc_device(img, psf, max_iterations=1, out=out, rng_seed=1)
    return old


print("round A: one workspace at a time", flush=True)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rl.clear_rlgc_caches(clear_memory_pool=True)
    rl.rlgc_device(img, psf, max_iterations=1, out=out, rng_seed=1)
    print(f"  trial {t:2d}: {timed():.3f} {timed():.3f} ms per iteration", flush=True)

print("round B: four held at once", flush=True)
for rep in range(3):
    rl.clear_rlgc_caches(clear_memory_pool=True)
    held, times = [], []
    for k in range(4):
        held += fresh_plan()  # keeps the previous plans (and their workspaces) alive
        plan = next(iter(rl._plan_cache.values()))
        a, b = timed(), timed()
        times.append((min(a, b), k, plan))
        print(f"  rep {rep} candidate {k}: {a:.3f} {b:.3f}", flush=True)
    best = min(times, key=lambda v: v[0])
    key = next(iter(rl._plan_cache.keys()))
    cur = rl._plan_cache[key]
    if cur is not best[2]:
        held.append(cur)
        rl._plan_cache[key] = best[2]
    lib = rl._hip.lib()
    for p in held:
        if p is not best[2]:
            lib.m3d_rlgc_plan_destroy(p.handle)
    held.clear()
    times.clear()
    del cur, plan
    torch.cuda.empty_cache()
    print(f"  rep {rep}: kept candidate {best[1]} ({best[0]:.3f}); alone again: {timed():.3f} {timed():.3f}"
          f"  (free {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB)", flush=True)
