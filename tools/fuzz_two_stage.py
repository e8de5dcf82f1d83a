"""Many more seeds of the box / parallelogram / mixed-soup parity scenes than the test suite runs (tests/test_gpu_parity.py
makes the scenes), two-stage kernel against the CPU oracle, both sample modes: 240 cases, about three minutes on the GPU box.

    python3 tools/fuzz_two_stage.py [first_seed [n_seeds]]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("opencl-path-tracer_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import bdpt, oracle_bind as oracle
import test_gpu_parity as T
t0 = time.time()
n = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + count):
    for make, args in ((T._boxes, (7,)), (T._quad_walls, (34, 16)), (T._mixed_soup, (110,))):
        scene = make(bdpt, *args, seed)
        for mode in (dict(), dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)):
            W, H = 48, 32
            ctx, st, p = T._pair(bdpt, oracle, scene, W, H, seed=seed, kernel_variant=bdpt.KERNEL_LDS2, **mode)
            for s in range(2):
                ctx.render(5)
                oracle.render(st, scene, scene.camera, p, 5, n_threads=16)
                T._compare(bdpt, ctx, st, "%s seed %d mode %s launch %d:" % (make.__name__, seed, mode, s))
            ctx.close()
            n += 1
    print("seed", seed, "ok", "%.0f s" % (time.time() - t0), flush=True)
print("all", n, "cases bit-exact")
