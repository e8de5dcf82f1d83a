/*
 * bdpt_class_pool.h -- index-order launches with a wavefront's pixels sorted by the length of their next sample
 * (included by bdpt_kernels.hip).
 *
 * A lane that owns one pixel walks every sample in lock-step with its wavefront, which pays for the longest light
 * subpath of its 64 samples: ten searches per sample where the lanes need seven on average.  STRIDED + SUM launches
 * hide that by taking each pixel's samples class by class (SampleCursor, bdpt_device.h); the reference's own mode --
 * sequential generator, running mean (kernel0.cl:590, :853-863) -- cannot reorder a pixel's samples.  But the number of
 * vertices Russian roulette allows a sample (kernel0.cl:787-788) follows from the generator word the sample starts
 * with, in that mode too: its roulette draws are the generator's outputs number 9, 13, 17, ... of the sample whatever
 * the rays hit.  So the pixels can be sorted instead of the samples.
 *
 * A wavefront owns BDPT_CPOOL_W x BDPT_CPOOL_ROWS pixels (16 x 8 = 128, two per lane).  Each pixel carries the class of
 * its NEXT sample and the number of samples it has taken (one word of LDS) and its generator word (another).  Every
 * round the wavefront counts the pixels by class with ballots and chooses a WINDOW of classes hi >= c >= lo: filling
 * the 64 lanes from class hi downwards, the window whose samples waste the fewest lane-steps (sum of the members'
 * classes over 64 x hi is largest).  A prefix count over the ballots hands the members their lanes -- the scan start
 * moves round the lanes from round to round so that no pixel is left behind -- and the wavefront runs ONE WHOLE SAMPLE
 * for them, the same sample_direct() the per-pixel kernels run.  A pixel's samples are taken in index order, one after
 * the other, with the reference's arithmetic per sample (:587-590, :853-863), so every bit of the result is that of
 * the per-pixel kernels; only the order in which different pixels advance differs.
 *
 * The accumulated colour stays in the image: a sample asks for its pixel's colour when it starts, needs it ten searches
 * later and stores the new one (kernel0.cl:587, :861-863), write-through when another run of the launch will read it.
 * A run of a sliced launch is a share of the tile's WORK, not of every pixel's samples: pixels are left wherever they
 * are when the budget is spent (generator word and progress go to memory then), and only the last run drains the
 * pool.  Nothing is allocated: the kernel needs the image, the seeds and the sliced launches' cursor array.
 *
 * The selection costs some hundred instructions per sample of 6 000; nothing of a path's state leaves the registers.
 */
#ifndef BDPT_CLASS_POOL_H
#define BDPT_CLASS_POOL_H

namespace bdpt {

#define BDPT_CPOOL_SLOTS (BDPT_CPOOL_W * BDPT_CPOOL_ROWS)
#define BDPT_CPOOL_PER_LANE (BDPT_CPOOL_SLOTS / 64)
/* LDS of one wavefront, in float4: two words per slot (a lane's slots side by side: their class words, then their generator
 * words), the 64 slot numbers of a round */
#define BDPT_CPOOL_LANE_WORDS (2 * BDPT_CPOOL_PER_LANE)
#define BDPT_CPOOL_WAVE_F4 ((64 * BDPT_CPOOL_LANE_WORDS + 3) / 4 + 16)
#define BDPT_CPOOL_BLOCK_MAX 256          /* threads per workgroup, at most */
#define BDPT_CPOOL_MAX_CLASS 5            /* launches with longer light subpaths run one lane per pixel */

static_assert(BDPT_CPOOL_SLOTS % 64 == 0 && BDPT_CPOOL_PER_LANE <= 8, "a lane looks after BDPT_CPOOL_PER_LANE slots");

/* slot word: class of the pixel's next sample (bits 0-7; 0 = the pixel has taken all samples of the launch) | samples of the
 * launch it has taken << 8 (hence launches of fewer than 2^24 samples per pixel) */

template <bool AOV, bool COUNT, int MODE, bool ALT>
__global__ __launch_bounds__(BDPT_CPOOL_BLOCK_MAX) __attribute__((amdgpu_waves_per_eu(MODE == MODE_LDS2 ? BDPT_LDS2_WAVES : MODE == MODE_BVH ? BDPT_BVH_WAVES : 6, MODE == MODE_LDS2 ? BDPT_LDS2_WAVES : MODE == MODE_BVH ? BDPT_BVH_WAVES : 6)))
void bdpt_class_pool_kernel(const bdpt_kargs K)
{
    constexpr int PER = BDPT_CPOOL_PER_LANE;
    constexpr int P = BDPT_CPOOL_SLOTS;
    const bdpt_dev_scene &S = K.scene;
    if (MODE == MODE_LDS) lds_fill_scene<false>(S);          /* every thread of the workgroup, before any early exit */
    if (MODE == MODE_LDS2) lds_fill_scene<true>(S);
    const Slice sl = slice_begin(K);
    if (sl.abandoned) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = K.block >> 6;
    /* behind the scene and the lanes' parked eye vertices (lane_park), or behind the stacks of the hierarchy walk */
    const LdsLayout L = MODE == MODE_LDS2 ? lds_layout<true>(S) : lds_layout<false>(S);
    const int base_f4 = MODE == MODE_DIRECT ? 0 : MODE == MODE_BVH ? (int)(blockDim.x >> 6) * (S.bvh_stack_limit * 64 / 4 + BDPT_BVH_SHARE_BYTES / 16)     /* behind the walk's stacks */
                                                : L.ids + L.n_ids / 4 + (int)blockDim.x * BDPT_PARK_F4;
    uint32_t *const words = (uint32_t *)(bdpt_lds + base_f4 + wv * BDPT_CPOOL_WAVE_F4);      /* [64][BDPT_CPOOL_LANE_WORDS] */
    uint32_t *const list = words + (64 * BDPT_CPOOL_LANE_WORDS + 3) / 4 * 4;
    constexpr int LW = BDPT_CPOOL_LANE_WORDS;
    const bool strided = K.rng_mode == BDPT_RNG_STRIDED;
    const bool last_run = sl.run + 1 >= K.n_chunks;

    /* this wavefront's pixels, the wavefronts of a workgroup side by side; slot s = pixel (s % BDPT_CPOOL_W, s / BDPT_CPOOL_W) of them */
    const int x0 = K.rx0 + ((sl.tile % K.wg_x) * nw + wv) * BDPT_CPOOL_W, y0 = K.ry0 + (sl.tile / K.wg_x) * BDPT_CPOOL_ROWS;
    auto class_of = [&](uint32_t start_state) {
        return (uint32_t)(K.max_l <= 5 ? roulette_class<4>(K, start_state) : K.max_l <= 7 ? roulette_class<6>(K, start_state) : roulette_class<0>(K, start_state));
    };

    /* A run is a share of the tile's work, not of every pixel's: it runs rounds until it has taken as many samples as
     * the tile has pixels times chunk_spp, and leaves every pixel wherever it is (its count in K.slice_cursor); only the
     * last run takes every pixel to the end, so the pool runs empty once per launch, not once per run. */
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int slot = lane * PER + j;
        const int x = x0 + slot % BDPT_CPOOL_W, y = y0 + slot / BDPT_CPOOL_W;
        uint32_t word = 0u;
        if (x < K.rx0 + K.rw && y < K.ry0 + K.rh) {
            const size_t pix = (size_t)y * K.width + x;
            const uint32_t taken = sl.run == 0 ? 0u : K.slice_cursor[pix].x;
            /* the generator word the pixel's next sample starts with (kernel0.cl:590; strided launches: the seed
             * advanced by whole samples, and never written back) */
            uint32_t start = K.seed_w[pix];
            if (strided) start *= pow_u32(kSampleStride, K.first_sample + taken);
            words[lane * LW + PER + j] = start;
            word = ((int)taken < K.spp ? class_of(start) : 0u) | (taken << 8);
        }
        words[lane * LW + j] = word;
    }
    long long budget = (long long)P * K.chunk_spp;

    Tally T = {};
    int rot = 0;
    while (last_run || budget > 0) {
        /* ---- which class, which pixels ---- */
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const int from = (lane + rot) & 63;                  /* the lane whose slots this lane looks at in this round */
        uint32_t cls[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) cls[j] = words[from * LW + j] & 0xFFu;
        /* members per class, then the window of classes hi >= c >= lo, taken from the top, that wastes least: its 64
         * samples (or fewer) are traced in lock-step for as long as the longest of them, class hi, takes */
        int cnt[BDPT_CPOOL_MAX_CLASS + 1];
#pragma unroll
        for (int c = 1; c <= BDPT_CPOOL_MAX_CLASS; c++) {
            int n = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) n += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls[j] == (uint32_t)c));
            cnt[c] = n;
        }
        int hi = 0, lo = 0, best_tot = 0, best_fill = 0;
#pragma unroll
        for (int h = BDPT_CPOOL_MAX_CLASS; h >= 1; h--) {
            if (cnt[h] == 0) continue;
            int n = 0, tot = 0, l = h;
#pragma unroll
            for (int c = h; c >= 1; c--) {
                const int take = min(cnt[c], 64 - n);
                n += take;
                tot += take * c;
                l = take > 0 ? c : l;
            }
            if (hi == 0 || tot * hi > best_tot * h) { hi = h; lo = l; best_tot = tot; best_fill = n; }    /* tot / h > best_tot / hi */
        }
        if (hi == 0) break;
        int before = 0;                                       /* members in front of this lane's j-th slot */
#pragma unroll
        for (int c = BDPT_CPOOL_MAX_CLASS; c >= 1; c--) {
            if (c > hi || c < lo) continue;
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const bool m = cls[j] == (uint32_t)c;
                const unsigned long long b = __builtin_amdgcn_ballot_w64(m);
                const int rank = before + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (m && rank < 64) list[rank] = (uint32_t)(from * PER + j);
                before += __builtin_popcountll(b);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        rot = (rot + 17) & 63;
        budget -= best_fill;
        if (lane < best_fill) {
            /* ---- one sample of one pixel (the body of direct_pixel's loop) ---- */
            const int slot = (int)list[lane];                /* all that has to survive the sample: the rest is read again after it */
            uint32_t w, draws = 0, rays = 0, hash = 0;
            int eye_id[2] = {-1, -1};
            float col[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const float4 a4 = K.accum[(size_t)(y0 + slot / BDPT_CPOOL_W) * K.width + (x0 + slot % BDPT_CPOOL_W)];     /* kernel0.cl:587; asked for now, needed when the sample is done */
            bool early;
            {
                const int px = x0 + slot % BDPT_CPOOL_W, py = y0 + slot / BDPT_CPOOL_W;
                const uint32_t *const wp = &words[(slot / PER) * LW + slot % PER];
                w = wp[PER];
                if (AOV) { const size_t pix = (size_t)py * K.width + px; draws = K.aov_rng_draws[pix]; rays = K.aov_rays[pix]; hash = K.aov_hit_hash[pix]; }
                float xt, yt;
                pixel_coord(K, px, py, xt, yt);                               /* :600-604 */
                const uint32_t rays_before = rays;
                early = sample_direct<AOV, MODE, ALT, AOV, COUNT>(K, xt, yt, w, draws, rays, hash, eye_id, col, T);
                if (COUNT) { T.rays += rays - rays_before; if (early) T.early++; else T.full++; }
            }
            asm volatile("" ::: "memory");
            const int px = x0 + slot % BDPT_CPOOL_W, py = y0 + slot / BDPT_CPOOL_W;
            const size_t pix = (size_t)py * K.width + px;
            uint32_t *const wp = &words[(slot / PER) * LW + slot % PER];
            const int s = (int)(*wp >> 8);                   /* samples of the launch this pixel has taken */
            const int n = K.num_samples + s;                                  /* kernel0.cl:588 */
            const uint32_t w_before = wp[PER];
            float acc[4] = {a4.x, a4.y, a4.z, a4.w};
            int taken = s + 1;
            if (early && !strided) {
                /* :645-656: raw value written, history dropped, the seed not written back: every later sample of the
                 * launch repeats this one exactly (as in direct_pixel).  The pixel is done. */
                w = w_before;
                acc[0] = col[0]; acc[1] = col[1]; acc[2] = col[2]; acc[3] = col[3];
                const int rest = K.spp - 1 - s;
                if (AOV) {
                    const uint32_t id = hit_id(eye_id[0], eye_id[1]);
                    for (int r = 0; r < rest; r++) hash = hash_step(hash, id);
                    draws += 3u * (uint32_t)rest;
                    rays += (uint32_t)rest;
                }
                if (COUNT) T.early += (unsigned long long)rest;
                taken = K.spp;
            } else if (n == 0) {                                              /* :855 */
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = col[k];
            } else if (K.accum_mode == BDPT_ACCUM_SUM) {
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = acc[k] + col[k];
            } else {                                                          /* :854 */
                const float s1 = clm_div((float)n, (float)(n + 1));
                const float s2 = clm_div(1.0f, (float)(n + 1));
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = clm_fma(col[k], s2, acc[k] * s1);
            }
            state_store(sl.shared, &K.accum[pix], make_float4(acc[0], acc[1], acc[2], acc[3]));      /* :861-863 */
            const uint32_t start = strided ? w_before * kSampleStride : w;    /* of the pixel's next sample */
            wp[PER] = start;
            if (AOV) {
                state_store(sl.shared, (uint2 *)&K.aov_eye_hit[pix * 2], make_uint2((uint32_t)eye_id[0], (uint32_t)eye_id[1]));
                state_store(sl.shared, &K.aov_hit_hash[pix], hash);
                state_store(sl.shared, &K.aov_radiance[pix], make_float4(col[0], col[1], col[2], col[3]));
                state_store(sl.shared, &K.aov_rng_draws[pix], draws);
                state_store(sl.shared, &K.aov_rays[pix], rays);
            }
            const uint32_t next = taken < K.spp ? class_of(start) : 0u;
            *wp = next | ((uint32_t)taken << 8);
        }
    }
    /* what the run leaves for the next run or the next launch (kernel0.cl:861-863) */
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int slot = lane * PER + j;
        const int x = x0 + slot % BDPT_CPOOL_W, y = y0 + slot / BDPT_CPOOL_W;
        if (x < K.rx0 + K.rw && y < K.ry0 + K.rh) {
            const size_t pix = (size_t)y * K.width + x;
            if (!strided) state_store(sl.shared, &K.seed_w[pix], words[lane * LW + PER + j]);
            if (!last_run) state_store(sl.shared, &K.slice_cursor[pix], make_uint4(words[lane * LW + j] >> 8, 0u, 0u, 0u));
        }
    }
    if (COUNT) {
        atomicAdd(&K.counters[0], T.rays);
        atomicAdd(&K.counters[1], T.full);
        atomicAdd(&K.counters[2], T.early);
        atomicAdd(&K.counters[3], T.cull);
        atomicAdd(&K.counters[4], T.exact);
    }
    slice_end(K, sl);
}

}  // namespace bdpt

#endif
