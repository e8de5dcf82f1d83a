/*
 * bdpt_pool.h -- the sample loop with the paths of a wavefront sorted by phase (included by bdpt_kernels.hip).
 *
 * In the per-pixel kernels a lane owns one pixel and walks its sample from the eye ray to the end of the light
 * subpath (kernel0.cl:592-847) in lock-step with the 63 other lanes of its wavefront: a lane whose subpath Russian
 * roulette ends early (:788), or whose bounce ray leaves the scene or hits a light (:796-813), idles until the longest
 * subpath of the wavefront is done.  In the reference's own mode (sequential generator, running mean) the samples of
 * a pixel cannot be reordered to hide that, and lane utilisation is 0.58.
 *
 * Here a wavefront owns a pool of more pixels than it has lanes (16 x BDPT_POOL_ROWS).  A pixel has at most one sample
 * in flight, so the order of its samples, its generator stream and its running mean are exactly those of the per-pixel
 * kernels; what changes is which lane works on it.  A sample is cut into phases that each end with one closest-hit
 * search:
 *
 *   EYE      (finish the pixel's previous sample: colour, kernel0.cl:846, blend, :853-856;) next sample, camera ray
 *            (:592-632), search, then :642-656 (ends at the eye ray) or the eye vertex and the light point (:669)
 *   CONNECT  connection ray from the current light-subpath vertex to the eye vertex (:699-705), search, connection
 *            term (:712-744), roulette (:787-788)
 *   BOUNCE   bounce direction (:767), search, :796-832
 *
 * Every round the wavefront counts the pool's pixels by phase with ballots, takes the phase most pixels wait in,
 * compacts up to 64 of them into its lanes (a prefix count over the ballots, slot numbers handed over through LDS;
 * the scan starts at a slot that moves round the pool so that no pixel is left behind), and runs that phase for all
 * of them together: one search per round, with the code before and after it the same for every lane.  Paths that have
 * ended simply wait in phase EYE; no lane idles for them.
 *
 * Where a path's state waits between rounds.  LDS is what bounds the number of wavefronts on a compute unit, so only
 * what a phase needs BEFORE its search (to form the ray) is kept there, three float4 per slot, one array per field so
 * that neighbouring slots are neighbouring addresses:
 *   H0  (current light-subpath vertex position, id of its primitive)     :675, :832
 *   H1  (its normal, generator word)
 *   H2  (eye vertex position, misc: eye surface, vertex number lti, what waits to be blended)
 * plus one word per slot: phase, and the number of samples the pixel has completed in this launch.  What is needed only
 * AFTER the search waits in device memory, four float4 per slot, in a region of K.pool_state that belongs to this
 * run of this tile alone (so no address is written from two dies during a launch and plain cached accesses are safe);
 * its loads are issued before the search and their latency disappears behind it:
 *   G0  (eye vertex normal, id of its primitive)                         :634
 *   G1  fullContrib (:689) -- or the raw colour of a sample that ended at the eye ray
 *   G2  accContrib (:692)
 *   G3  the pixel's accumulated colour (:587, :863): read from the image when the run starts, blended into here sample by
 *       sample, written to the image when the pixel has no more samples in this run
 */
#ifndef BDPT_POOL_H
#define BDPT_POOL_H

namespace bdpt {

#define BDPT_POOL_SLOTS (BDPT_POOL_W * BDPT_POOL_ROWS)
#define BDPT_POOL_LDS_F4 3
/* LDS of one wavefront, in float4: the slots, one phase word per slot, the 64 slot numbers of a round */
#define BDPT_POOL_WAVE_F4 (BDPT_POOL_SLOTS * BDPT_POOL_LDS_F4 + BDPT_POOL_SLOTS / 4 + 16)

static_assert(BDPT_POOL_SLOTS > 64 && BDPT_POOL_SLOTS <= 128 && BDPT_POOL_SLOTS % 4 == 0, "a lane looks after two slots at most");

/* (phase & 3) == 0: the slot's next ray is an eye ray */
enum { PH_EYE = 0, PH_CONNECT = 1, PH_BOUNCE = 2, PH_DONE = 3, PH_PROBE = 4 };

/* misc word of a slot: eye surface (bits 0-19), light-subpath vertex number (20-27), what waits to be blended into
 * the pixel (28-29: 0 nothing, 1 the path whose sums are in the slot, 2 the raw colour in G1) */
#define BDPT_POOL_SURF_MASK 0xFFFFFu
#define BDPT_POOL_LTI_SHIFT 20
#define BDPT_POOL_RES_SHIFT 28

BDPT_DEV uint32_t pack_id(int type, int idx) { return type < 0 ? 0u : (((uint32_t)(type + 1) << 28) | (uint32_t)idx); }
BDPT_DEV void unpack_id(uint32_t v, int &type, int &idx)
{
    type = (int)(v >> 28) - 1;
    idx = type < 0 ? -1 : (int)(v & 0x0FFFFFFFu);
}

/* multiplicative inverse mod 2^32 of an odd number (Newton: every step doubles the correct low bits) */
constexpr uint32_t inv_u32(uint32_t a)
{
    uint32_t x = a;
    for (int i = 0; i < 5; i++) x *= 2u - a * x;
    return x;
}
/* back from the generator word after the roulette draw to the word before the three draws of the cone sample */
constexpr uint32_t kBackFour = inv_u32(pow_u32_c(48271u, 4));
static_assert(kBackFour * pow_u32_c(48271u, 4) == 1u, "inverse of 48271^4");

BDPT_DEV float4 f4_bits(float x, float y, float z, uint32_t w) { return make_float4(x, y, z, __uint_as_float(w)); }

#ifndef BDPT_POOL_WAVES
#define BDPT_POOL_WAVES 6
#endif

template <bool AOV, bool COUNT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BDPT_POOL_WAVES, BDPT_POOL_WAVES)))
void bdpt_pool_kernel(const bdpt_kargs K)
{
    constexpr int P = BDPT_POOL_SLOTS;
    const bdpt_dev_scene &S = K.scene;
    if (MODE == MODE_LDS) lds_fill_scene<false>(S);          /* every thread of the workgroup, before any early exit */
    if (MODE == MODE_LDS2) lds_fill_scene<true>(S);
    const Slice sl = slice_begin(K);
    if (sl.abandoned) return;
    const LdsLayout L = MODE == MODE_LDS2 ? lds_layout<true>(S) : lds_layout<false>(S);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = K.block >> 6;
    float4 *const slots = bdpt_lds + L.ids + L.n_ids / 4 + wv * BDPT_POOL_WAVE_F4;
    uint32_t *const phase = (uint32_t *)(slots + P * BDPT_POOL_LDS_F4);
    uint32_t *const list = phase + P;
    /* this run's own region of device memory: no other workgroup of the launch touches it */
    float4 *const gstate = K.pool_state + ((size_t)(sl.run * K.n_tiles + sl.tile) * nw + wv) * (size_t)(P * BDPT_POOL_GLOBAL_F4);
#define HOT(f, s) slots[(f) * P + (s)]
#define COLD(f, s) gstate[(f) * P + (s)]
    const bool strided = K.rng_mode == BDPT_RNG_STRIDED;

    /* this wavefront's pixels: 16 x BDPT_POOL_ROWS, the wavefronts of a workgroup side by side */
    const int x0 = K.rx0 + ((sl.tile % K.wg_x) * nw + wv) * BDPT_POOL_W, y0 = K.ry0 + (sl.tile / K.wg_x) * BDPT_POOL_ROWS;

    for (int slot = lane; slot < P; slot += 64) {
        const int x = x0 + (slot & (BDPT_POOL_W - 1)), y = y0 + slot / BDPT_POOL_W;
        uint32_t phw = PH_DONE;
        if (x < K.rx0 + K.rw && y < K.ry0 + K.rh) {
            const size_t pix = (size_t)y * K.width + x;
            const uint32_t seed = K.seed_w[pix];
            HOT(1, slot) = f4_bits(0.0f, 0.0f, 0.0f, seed);
            HOT(2, slot) = f4_bits(0.0f, 0.0f, 0.0f, 0u);
            COLD(3, slot) = K.accum[pix];                                                                     /* :587 */
            bool probe = false;
            if (strided && sl.s0 == 0) {
                probe = K.reorder == 2;               /* ADAPTIVE order: the pixel's first eye ray decides, see PH_PROBE below */
                if (!probe) {
                    SampleCursor c;
                    cursor_init(c, K, seed);
                    if (K.class_masks) {
                        if (K.max_l <= 5) cursor_fill_masks<4>(K, sl.shared, pix, seed);
                        else cursor_fill_masks<6>(K, sl.shared, pix, seed);
                    }
                    state_store(sl.shared, &K.slice_cursor[pix], make_uint4((uint32_t)c.cls, (uint32_t)c.head, c.st_head, c.pending));
                }
            }
            phw = (probe ? PH_PROBE : PH_EYE) | ((uint32_t)sl.s0 << 3);
        }
        phase[slot] = phw;
    }

    Tally T = {};
    int rot = 0;
    for (;;) {
        /* ---- which phase, which slots ---- */
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        int sA = lane + rot, sB = lane + 64 + rot;
        sA -= sA >= P ? P : 0;
        sB -= sB >= P ? P : 0;
        const bool has_b = lane + 64 < P;
        const uint32_t pA = phase[sA] & 7u, pB = has_b ? phase[sB] & 7u : (uint32_t)PH_DONE;
        const int n_eye = __builtin_popcountll(__builtin_amdgcn_ballot_w64((pA & 3u) == PH_EYE)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64((pB & 3u) == PH_EYE));
        const int n_con = __builtin_popcountll(__builtin_amdgcn_ballot_w64(pA == PH_CONNECT)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(pB == PH_CONNECT));
        const int n_bnc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(pA == PH_BOUNCE)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(pB == PH_BOUNCE));
        if (n_eye + n_con + n_bnc == 0) break;
        const int k = (n_con >= n_bnc && n_con >= n_eye) ? PH_CONNECT : (n_bnc >= n_eye ? PH_BOUNCE : PH_EYE);
        const bool mA = (pA & 3u) == (uint32_t)k, mB = (pB & 3u) == (uint32_t)k;
        const unsigned long long bA = __builtin_amdgcn_ballot_w64(mA), bB = __builtin_amdgcn_ballot_w64(mB);
        const int rA = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bA, 0u));
        const int rB = __builtin_popcountll(bA) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bB, 0u));
        if (mA) list[rA] = (uint32_t)sA;                      /* rA < 64 */
        if (mB && rB < 64) list[rB] = (uint32_t)sB;
        const int n_run = min(64, __builtin_popcountll(bA) + __builtin_popcountll(bB));
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const bool active = lane < n_run;
        const int slot = active ? (int)list[lane] : 0;
        rot += 37;
        rot -= rot >= P ? P : 0;

        const int px = x0 + (slot & (BDPT_POOL_W - 1)), py = y0 + slot / BDPT_POOL_W;
        const size_t pix = active ? (size_t)py * K.width + px : 0;

        /* ---- the phase's part before its search: the ray, and the loads of what the part after it wants ---- */
        clm_f3 o, d;                                          /* set by every lane that shoots */
        int otype = 4, oidx = 0;
        bool shoot = false;
        float4 g0, g1, g2;                                    /* loaded by the phases whose second part reads them */
        if (active) {
            if (k == PH_EYE) {
                const uint32_t phw = phase[slot];
                uint32_t s = phw >> 3;
                const bool probe = (phw & 7u) == PH_PROBE;
                uint32_t misc = __float_as_uint(HOT(2, slot).w);
                const uint32_t res = misc >> BDPT_POOL_RES_SHIFT;
                float4 done4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     /* the pixel's colour, should this be its last visit */
                if (res != 0u) {
                    /* the pixel's previous sample: its colour (kernel0.cl:844-846) into the pixel (:853-863) */
                    const float4 fl = COLD(1, slot);
                    const float4 a4 = COLD(3, slot);
                    float col[4] = {fl.x, fl.y, fl.z, fl.w};
                    if (res == 1u) {
                        const int taken = (int)((misc >> BDPT_POOL_LTI_SHIFT) & 0xFFu) + 1;                  /* :757 */
                        const float inv_taken = clm_div(1.0f, (float)taken);                                  /* :846 */
                        const float4 ed4 = surface_row<MODE>(S, (misc & BDPT_POOL_SURF_MASK) * 2);
                        const float ed[4] = {ed4.x, ed4.y, ed4.z, ed4.w};
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            float v = K.light_emissive[c] * col[c] * inv_taken;
                            if (!K.indirect_only) v = v * ed[c];                                              /* :844 / :846 */
                            col[c] = v;
                        }
                    }
                    const int n = K.num_samples + (int)s;                                                     /* :588 */
                    float acc[4] = {a4.x, a4.y, a4.z, a4.w};
                    if (n == 0) {                                                                             /* :855 */
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = col[c];
                    } else if (K.accum_mode == BDPT_ACCUM_SUM) {
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = acc[c] + col[c];
                    } else {                                                                                  /* :854 */
                        const float s1 = clm_div((float)n, (float)(n + 1));
                        const float s2 = clm_div(1.0f, (float)(n + 1));
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = clm_fma(col[c], s2, acc[c] * s1);
                    }
                    done4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    COLD(3, slot) = done4;
                    if (AOV) state_store(sl.shared, &K.aov_radiance[pix], make_float4(col[0], col[1], col[2], col[3]));
                    s++;
                    misc &= ~(3u << BDPT_POOL_RES_SHIFT);
                    HOT(2, slot).w = __uint_as_float(misc);
                }
                /* the next sample, if the launch has one more for this pixel */
                uint32_t w = __float_as_uint(HOT(1, slot).w);
                bool more = probe || (int)s < sl.s1;
                if (strided) {
                    if (probe) w = cursor_first_state(K, K.seed_w[pix]);
                    else if (more) {
                        SampleCursor c;
                        cursor_load(c, K, pix);
                        uint32_t start = 0u;
                        more = sample_next(c, K, pix, start);
                        state_store(sl.shared, &K.slice_cursor[pix], make_uint4((uint32_t)c.cls, (uint32_t)c.head, c.st_head, c.pending));
                        w = start;
                    }
                    if (more && !probe) HOT(1, slot).w = __uint_as_float(w);
                }
                if (!more) {
                    if (res == 0u) done4 = COLD(3, slot);
                    state_store(sl.shared, &K.accum[pix], done4);                                             /* :863 */
                    if (!strided) state_store(sl.shared, &K.seed_w[pix], w);                                  /* :861 */
                    phase[slot] = PH_DONE | (s << 3);
                } else {
                    phase[slot] = (phw & 7u) | (s << 3);
                    uint32_t draws = 0;
                    const float xt = K.xt_tab[px], yt = K.yt_tab[py];                                         /* :600-604, tabulated by the host */
                    rng_float(w, draws);                                                                      /* :592, discarded */
                    d = camera_ray(w, draws, K, xt, yt);
                    o = clm_mad3(0.00001f, d, f3(K.cam_pos));                                                 /* :632 */
                    shoot = true;
                }
            } else if (k == PH_CONNECT) {
                g0 = COLD(0, slot); g1 = COLD(1, slot); g2 = COLD(2, slot);
                const float4 v0 = HOT(0, slot), e0 = HOT(2, slot);
                const int lti = (int)((__float_as_uint(e0.w) >> BDPT_POOL_LTI_SHIFT) & 0xFFu);
                o = clm_make3(v0.x, v0.y, v0.z);
                unpack_id(__float_as_uint(v0.w), otype, oidx);
                d = clm_normalize3(clm_sub3(clm_make3(e0.x, e0.y, e0.z), o));                                 /* :699 */
                shoot = AOV || !K.indirect_only || lti != 0;        /* INDIRECT_ONLY drops the light's own connection (:752-755) */
            } else {
                g2 = COLD(2, slot);
                const float4 v0 = HOT(0, slot), v1 = HOT(1, slot);
                const clm_f3 vpos = clm_make3(v0.x, v0.y, v0.z), vn = clm_make3(v1.x, v1.y, v1.z);
                unpack_id(__float_as_uint(v0.w), otype, oidx);
                o = clm_mad3(0.00001f, vn, vpos);                                                             /* :761 */
                uint32_t w_dir = __float_as_uint(v1.w) * kBackFour, draws_dir = 0;    /* the three numbers roulette stepped over */
                d = sample_cone(w_dir, draws_dir, vn, 3.1415f * 0.50f);                                       /* :767, :486 */
                shoot = true;
            }
        }

        /* ---- the search ---- */
        asm volatile("" ::: "memory");
        Hit h;
        if (shoot) h = trace_one<MODE, true, COUNT>(S, o, d, otype, oidx, T);
        asm volatile("" ::: "memory");

        /* ---- the phase's part after its search ---- */
        if (active) {
            if (k == PH_EYE) {
                if (shoot) {
                    const uint32_t phw = phase[slot];
                    const uint32_t s = phw >> 3;
                    const bool early_miss = h.type < 0;
                    float4 diffuse = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    bool early = early_miss;
                    if (!early_miss) {
                        diffuse = surface_row<MODE>(S, h.surf * 2);
                        early = surface_row<MODE>(S, h.surf * 2 + 1).w != 0.0f;                               /* :645 */
                    }
                    if ((phw & 7u) == PH_PROBE) {
                        /* ADAPTIVE order: class order for the pixel if this ray, the eye ray of the launch's first sample
                         * index, enters the light loop (as in direct_pixel) */
                        const uint32_t seed = K.seed_w[pix];
                        SampleCursor c;
                        cursor_init(c, K, seed);
                        c.cls = early ? -1 : c.cls;
                        if (K.class_masks && c.cls >= 1) {
                            if (K.max_l <= 5) cursor_fill_masks<4>(K, sl.shared, pix, seed);
                            else cursor_fill_masks<6>(K, sl.shared, pix, seed);
                        }
                        state_store(sl.shared, &K.slice_cursor[pix], make_uint4((uint32_t)c.cls, (uint32_t)c.head, c.st_head, c.pending));
                        phase[slot] = PH_EYE | (s << 3);
                        if (COUNT && !AOV) T.rays++;                 /* a ray this launch form traces on top of the samples' own */
                    } else {
                        if (COUNT) T.rays++;
                        uint32_t hash = 0u;
                        if (AOV) {
                            hash = hash_step(K.aov_hit_hash[pix], hit_id(h.type, h.idx));
                            state_store(sl.shared, (uint2 *)&K.aov_eye_hit[pix * 2], make_uint2((uint32_t)h.type, (uint32_t)h.idx));
                        }
                        if (early) {                                                                          /* :645-656 */
                            const float col[4] = {early_miss ? 0.0f : diffuse.x, early_miss ? 0.0f : diffuse.y,
                                                  early_miss ? 0.0f : diffuse.z, early_miss ? 0.0f : diffuse.w};
                            if (!strided) {
                                /* raw value written, history dropped, the seed not written back: every later sample of
                                 * this launch repeats this one exactly (as in direct_pixel).  The pixel is done. */
                                const int rest = sl.s1 - 1 - (int)s;
                                state_store(sl.shared, &K.accum[pix], make_float4(col[0], col[1], col[2], col[3]));
                                state_store(sl.shared, &K.seed_w[pix], __float_as_uint(HOT(1, slot).w));
                                if (AOV) {
                                    const uint32_t id = hit_id(h.type, h.idx);
                                    for (int r = 0; r < rest; r++) hash = hash_step(hash, id);
                                    state_store(sl.shared, &K.aov_hit_hash[pix], hash);
                                    state_store(sl.shared, &K.aov_rng_draws[pix], K.aov_rng_draws[pix] + 3u + 3u * (uint32_t)rest);
                                    state_store(sl.shared, &K.aov_rays[pix], K.aov_rays[pix] + 1u + (uint32_t)rest);
                                    state_store(sl.shared, &K.aov_radiance[pix], make_float4(col[0], col[1], col[2], col[3]));
                                }
                                if (COUNT) T.early += 1ull + (unsigned long long)rest;
                                phase[slot] = PH_DONE | (s << 3);
                            } else {
                                if (AOV) {
                                    state_store(sl.shared, &K.aov_hit_hash[pix], hash);
                                    state_store(sl.shared, &K.aov_rng_draws[pix], K.aov_rng_draws[pix] + 3u);
                                    state_store(sl.shared, &K.aov_rays[pix], K.aov_rays[pix] + 1u);
                                }
                                if (COUNT) T.early++;
                                COLD(1, slot) = make_float4(col[0], col[1], col[2], col[3]);
                                HOT(2, slot).w = __uint_as_float(2u << BDPT_POOL_RES_SHIFT);
                            }
                        } else {
                            if (COUNT) T.full++;
                            uint32_t w = __float_as_uint(HOT(1, slot).w) * lcg_pow(3), draws = 0;      /* after :592 and the two of :613 */
                            clm_f3 vpos, vn;
                            sample_light(w, draws, K, vpos, vn);                                              /* :669 */
                            HOT(0, slot) = f4_bits(vpos.x, vpos.y, vpos.z, pack_id(K.light_origin_type, K.light_origin_index));
                            HOT(1, slot) = f4_bits(vn.x, vn.y, vn.z, w);
                            HOT(2, slot) = f4_bits(h.pos.x, h.pos.y, h.pos.z, h.surf & BDPT_POOL_SURF_MASK);  /* lti = 0, nothing to blend */
                            COLD(0, slot) = f4_bits(h.n.x, h.n.y, h.n.z, pack_id(h.type, h.idx));
                            /* fullContrib = 0 (:689) and accContrib = 1 (:692) are what CONNECT and BOUNCE take at lti = 0 */
                            phase[slot] = PH_CONNECT | (s << 3);
                            if (AOV) {
                                state_store(sl.shared, &K.aov_hit_hash[pix], hash);
                                state_store(sl.shared, &K.aov_rng_draws[pix], K.aov_rng_draws[pix] + 5u);
                                state_store(sl.shared, &K.aov_rays[pix], K.aov_rays[pix] + 1u);
                            }
                        }
                    }
                }
            } else if (k == PH_CONNECT) {
                const float4 v1 = HOT(1, slot), e0 = HOT(2, slot);
                const uint32_t misc = __float_as_uint(e0.w);
                const int lti = (int)((misc >> BDPT_POOL_LTI_SHIFT) & 0xFFu);
                if (shoot) {
                    if (COUNT) T.rays++;
                    Hit at;
                    at.pos = clm_make3(e0.x, e0.y, e0.z);
                    unpack_id(__float_as_uint(g0.w), at.type, at.idx);
                    at.n = clm_make3(g0.x, g0.y, g0.z);
                    at.surf = 0;
                    float pre = K.pre_factor0;                                                                /* :694, :745 */
                    if (lti >= BDPT_INV_PRE_N)
                        for (int i = 0; i < lti; i++) pre *= K.p_con;
                    const bool first = lti == 0;
                    const float acc[4] = {first ? 1.0f : g2.x, first ? 1.0f : g2.y, first ? 1.0f : g2.z, first ? 1.0f : g2.w};   /* :692 */
                    float contrib[4];
                    connection(K, at, h, o, clm_make3(v1.x, v1.y, v1.z), d, acc, inv_pre_factor(K, lti, pre), contrib);
                    if (K.indirect_only && first) contrib[0] = contrib[1] = contrib[2] = contrib[3] = 0.0f;   /* :752-755 */
                    /* adding a zero changes nothing: only a first vertex (which starts the sum, :689) or a connection that
                     * carries something is stored */
                    if (first || contrib[0] != 0.0f || contrib[1] != 0.0f || contrib[2] != 0.0f || contrib[3] != 0.0f)
                        COLD(1, slot) = make_float4((first ? 0.0f : g1.x) + contrib[0], (first ? 0.0f : g1.y) + contrib[1],
                                                    (first ? 0.0f : g1.z) + contrib[2], (first ? 0.0f : g1.w) + contrib[3]);
                    if (AOV) {
                        state_store(sl.shared, &K.aov_hit_hash[pix], hash_step(K.aov_hit_hash[pix], hit_id(h.type, h.idx)));
                        state_store(sl.shared, &K.aov_rays[pix], K.aov_rays[pix] + 1u);
                    }
                } else if (lti == 0) {
                    COLD(1, slot) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                      /* :689 */
                }
                /* roulette (:787-788) over the three numbers of the bounce direction, which BOUNCE recovers */
                uint32_t w = __float_as_uint(v1.w) * lcg_pow(3), draws = 0;
                const float fate = rng_float(w, draws);
                HOT(1, slot).w = __uint_as_float(w);
                if (AOV) state_store(sl.shared, &K.aov_rng_draws[pix], K.aov_rng_draws[pix] + 4u);
                const bool ends = fate > K.p_con || (!AOV && lti + 1 == K.max_l);   /* nothing but the parity buffers sees the last bounce ray */
                if (ends) HOT(2, slot).w = __uint_as_float(misc | (1u << BDPT_POOL_RES_SHIFT));
                phase[slot] = (phase[slot] & ~7u) | (ends ? (uint32_t)PH_EYE : (uint32_t)PH_BOUNCE);
            } else {
                if (COUNT) T.rays++;
                if (AOV) {
                    state_store(sl.shared, &K.aov_hit_hash[pix], hash_step(K.aov_hit_hash[pix], hit_id(h.type, h.idx)));
                    state_store(sl.shared, &K.aov_rays[pix], K.aov_rays[pix] + 1u);
                }
                uint32_t misc = __float_as_uint(HOT(2, slot).w);
                const int lti = (int)((misc >> BDPT_POOL_LTI_SHIFT) & 0xFFu);
                bool ends = h.type < 0;                                                                       /* :812 */
                if (!ends) {
                    const float4 nd = surface_row<MODE>(S, h.surf * 2);
                    ends = surface_row<MODE>(S, h.surf * 2 + 1).w != 0.0f || lti + 1 >= K.max_l;              /* :797-803 */
                    if (!ends) {
                        const bool first = lti == 0;
                        COLD(2, slot) = make_float4((first ? 1.0f : g2.x) * nd.x, (first ? 1.0f : g2.y) * nd.y,
                                                    (first ? 1.0f : g2.z) * nd.z, (first ? 1.0f : g2.w) * nd.w);   /* :827 with :822, :824 */
                        HOT(0, slot) = f4_bits(h.pos.x, h.pos.y, h.pos.z, pack_id(h.type, h.idx));            /* :832 */
                        const float w_keep = HOT(1, slot).w;
                        HOT(1, slot) = make_float4(h.n.x, h.n.y, h.n.z, w_keep);
                        misc += 1u << BDPT_POOL_LTI_SHIFT;
                    }
                }
                if (ends) misc |= 1u << BDPT_POOL_RES_SHIFT;
                HOT(2, slot).w = __uint_as_float(misc);
                phase[slot] = (phase[slot] & ~7u) | (ends ? (uint32_t)PH_EYE : (uint32_t)PH_CONNECT);
            }
        }
    }
#undef HOT
#undef COLD
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
