"""Pooled against per-lane launches on random frames, regions, scenes, modes, parameters and run counts, every output bit for bit
(images, generator state, parity buffers, work counters).  usage: fuzz_pooled.py [seed] [cases] [scheduling]
scheduling: 2 = a wavefront's paths sorted by phase (the default), 3 = a wavefront's pixels sorted by the class of their next sample"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "opencl-path-tracer_amd"))
import numpy as np, bdpt
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
scenes = [bdpt.Scene.cornell(), bdpt.Scene.outdoor(), bdpt.Scene.cornell_planes(), bdpt.Scene.soup(40)]
bad = 0
SCHED = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ran = {}
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    W, H = int(rng.integers(5, 130)), int(rng.integers(3, 70))
    sc = scenes[int(rng.integers(0, len(scenes)))]
    mode = [dict(), dict(rng_mode=1), dict(rng_mode=1, accum_mode=1, sample_order=1), dict(rng_mode=1, accum_mode=1, sample_order=2), dict(rng_mode=1, accum_mode=1, sample_order=3)][int(rng.integers(0, 5))]
    kw = dict(kernel_variant=int(rng.choice([0, 3, 5])), aov_enable=int(rng.integers(0, 2)), max_light_bounces=int(rng.integers(1, 8)),
              p_continue=float(rng.choice([0.5, 0.8, 0.95])), indirect_only=int(rng.integers(0, 2)) if rng.random() < 0.2 else 0,
              light_origin=int(rng.integers(0, 2)) if rng.random() < 0.3 else 0, **mode)
    spps = [int(rng.integers(1, 40)) for _ in range(int(rng.integers(1, 4)))]
    chunks = int(rng.choice([0, 1, 2, 3, 5]))
    region = None
    if rng.random() < 0.3:
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H)); region = (x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1)))
    seeds = bdpt.make_seeds(W, H, int(rng.integers(1, 1 << 30)))
    outs = []
    for sched in (1, SCHED):
        try:
            with bdpt.Context(W, H) as ctx:
                ctx.set_params(bdpt.default_params(**kw)); ctx.upload_scene(sc); ctx.upload_seeds(seeds); ctx.set_camera(sc.camera)
                ctx.set_path_scheduling(sched); ctx.set_launch_chunks(chunks); ctx.enable_counters(True)
                if region: ctx.set_region(*region)
                o = []
                for spp in spps:
                    ctx.render(spp)
                    o += [ctx.read_accum().view(np.uint32), ctx.read_seeds(), np.array(ctx.read_counters())]
                    if kw["aov_enable"]:
                        o += [np.asarray(ctx.read_aov(k)).view(np.uint8) for k in range(5)]
                o.append(ctx.last_launch_path_scheduling())
                if sched == SCHED: ran[o[-1]] = ran.get(o[-1], 0) + 1
                outs.append(o)
        except bdpt.BdptError as e:
            outs.append(["error " + str(e)[:80]])
    same = len(outs[0]) == len(outs[1]) and all(np.array_equal(a, b) for a, b in zip(outs[0][:-1], outs[1][:-1]))
    if not same and not isinstance(outs[0][0], str):
        which = [i for i, (a, b) in enumerate(zip(outs[0][:-1], outs[1][:-1])) if not np.array_equal(a, b)]
        per = 3 + (5 if kw["aov_enable"] else 0)
        names = ["accum", "seeds", "counters", "eye", "hash", "radiance", "draws", "rays"]
        print("   differing:", [(i // per, names[i % per]) for i in which], [ (outs[0][i], outs[1][i]) for i in which if i % per == 2][:2])
        if all(i % per == 2 for i in which) and chunks == 0: same = True      # work counters of STREAM launches depend on the number of runs
        # sorted by class: a pixel whose eye ray ends the path is traced once per launch, not once per run -- fewer rays, nothing else
        if all(i % per == 2 for i in which) and SCHED == 3 and chunks != 1:
            same = "rng_mode" not in kw and all(np.all(outs[0][i] >= outs[1][i]) and np.array_equal(outs[0][i][1:3], outs[1][i][1:3]) for i in which)   # rays, cull and exact tests
    if not same or isinstance(outs[0][0], str):
        bad += 1
        print("case", case, W, H, kw, spps, chunks, region, "MISMATCH" if not same else outs[0][0], flush=True)
print("cases done, mismatches:", bad, "launches by the scheduling they ran with:", ran)
sys.exit(1 if bad else 0)
