// kernel_emu.cpp -- TEST ONLY.  Executes the real kernel sources (pyramid.hip, lk.hip) on the CPU
// through the coroutine SIMT emulator in hip_emu.h, so that the CPU test-suite can compare the
// kernels' complete data path (bordered pyramid layout, Scharr images, lane mapping, packed pixel
// arithmetic, DPP reductions, tile handling) with the oracle without a GPU.  Not a product path.
#include "hip_emu.h"

#include "../../visual_odom_amd/csrc/fast.hip"
#include "../../visual_odom_amd/csrc/lk.hip"
#include "../../visual_odom_amd/csrc/pyramid.hip"
#include "../../visual_odom_amd/csrc/post.hip"
#include "../../visual_odom_amd/csrc/pnp.hip" // (brings vo_epnp.h, vo_svd_wide.h, vo_p3p.h; host launch code is compiled out)
#include "../../visual_odom_amd/csrc/essential.hip"
#include "../../visual_odom_amd/csrc/seq.hip"

#include <memory>
#include <vector>

namespace {

struct Plan {
    int levels = 0;
    int lw[VO_MAX_LEVELS], lh[VO_MAX_LEVELS], ls[VO_MAX_LEVELS];
    size_t off[VO_MAX_LEVELS], total = 0;
};

// the geometry libvo_hip plans in capi.hip (plan_levels / level_stride)
Plan plan(int w, int h, int max_level)
{
    Plan p;
    int cw = w, ch = h, l = 0;
    size_t off = 0;
    for (;; l++) {
        p.lw[l] = cw;
        p.lh[l] = ch;
        p.ls[l] = (VO_BX + cw + VO_BY + 15) / 16 * 16;
        p.off[l] = off;
        off += (size_t)p.ls[l] * (ch + 2 * VO_BY);
        off = (off + 255) / 256 * 256;
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (l == max_level || l + 1 >= VO_MAX_LEVELS || nw <= 21 || nh <= 21)
            break;
        cw = nw;
        ch = nh;
    }
    p.levels = l + 1;
    p.total = off;
    return p;
}

// The image table of the emulated runs: every level of every image in its OWN heap block of exactly ls * (lh + 2 VO_BY)
// bytes / dwords.  That is tighter than the product's table (capi.hip: levels one after the other at 256-byte boundaries, the
// images one after the other, vo_create's worst-case slack behind the last): under AddressSanitizer (VO_SANITIZE=1,
// tests/test_sanitize.py) any kernel access outside a level's bordered allocation aborts, whichever level, image or pyramid
// depth it belongs to.  Pixels are poisoned with 0xA5 (a read of border the build did not write shows up), derivatives zero.
struct Heap {
    Plan p;
    std::vector<std::unique_ptr<uint8_t[]>> pix;
    std::vector<std::unique_ptr<uint32_t[]>> der;
    std::vector<vo::PyrImage> tab;
    size_t level_elems(int l) const { return (size_t)p.ls[l] * (p.lh[l] + 2 * VO_BY); }
    uint8_t *pix_block(int i, int l) { return pix[(size_t)i * p.levels + l].get(); }
    uint32_t *der_block(int i, int l) { return der[(size_t)i * p.levels + l].get(); }
    Heap(const Plan &plan_, int n_img, const uint8_t *imgs, int w, int h) : p(plan_), tab(n_img)
    {
        for (int i = 0; i < n_img; i++) {
            memset(&tab[i], 0, sizeof(vo::PyrImage));
            for (int l = 0; l < p.levels; l++) {
                const size_t n = level_elems(l), org = (size_t)VO_BY * p.ls[l] + VO_BX;
                pix.emplace_back(new uint8_t[n]);
                der.emplace_back(new uint32_t[n]);
                memset(pix.back().get(), 0xA5, n);
                memset(der.back().get(), 0, 4 * n);
                tab[i].lvl[l] = pix.back().get() + org;
                tab[i].der[l] = der.back().get() + org;
                tab[i].w[l] = p.lw[l];
                tab[i].h[l] = p.lh[l];
                tab[i].stride[l] = p.ls[l];
            }
            for (int y = 0; y < h; y++)
                memcpy(tab[i].lvl[0] + (ptrdiff_t)y * p.ls[0], imgs + ((size_t)i * h + y) * w, w);
        }
    }
};

int g_bucket_threads = 256; // ke_set_bucket_threads: width of bucket_kernel's workgroup (256, or 1024 as in launches of <= 4 frames)
int g_fast_big = 0; // ke_set_fast_big: tile form of the FAST kernel (0: 64 x 16, 1: 64 x 32, 2: 128 x 32)
int g_lk_pair = 0; // ke_set_lk_pair: run the two-features-per-wavefront LK kernel instead (1) / the split chain (2)
bool g_lk_split_fine = false;

template <typename F>
void launch(unsigned gx, unsigned gy, unsigned gz, int threads, F body)
{
    for (unsigned z = 0; z < gz; z++)
        for (unsigned y = 0; y < gy; y++)
            for (unsigned x = 0; x < gx; x++)
                emu::run_block(threads, x, y, z, body);
}

int g_pyr_lds = 2; // (default: the product chain) ke_set_pyr_lds: which pyramid chain build_pyramids emulates (0 / 1: three kernels with the column-walk / LDS pyr_down; 2: fused passes)

// the PYRAMID stage of capi.hip's run_stages with the emulated kernels
void build_pyramids(const Plan &p, const vo::PyrImage *d_imgs, int n_img)
{
    using namespace vo;
    if (g_pyr_lds == 2 || g_pyr_lds == 3) { // the fused passes (3: workgroups in dispatch order, what launches of fewer than 16 images use)
        const int remap = g_pyr_lds == 2;
        // the fused passes (round 4; what launch_pyramid_fused enqueues): one launch per level
        const PassPlan pp = pass_plan(p.levels, p.lw, p.lh, p.ls, /*wide border items*/ remap != 0); // (the product's two regimes: many images =
                                                                                                       //  XCD-pinned order + wide items, a few = dispatch order + thin)
        for (int l = 0; l < p.levels; l++)
            {
                const uint32_t nwg = pass_grid(pp, l, (int)n_img, remap);
                launch(nwg, 1, 1, 64, [&] { pyr_pass_kernel(d_imgs, l, p.levels, pp, (uint32_t)n_img, remap); });
            }
        return;
    }
    // the three-kernel chain: level 0's border + Scharr image, the pyr_down chain, then the other levels
    for (int first = 0, last = 1; first < p.levels; first = last, last = p.levels) {
        if (first == 1)
            for (int l = 0; l + 1 < p.levels; l++)
                if (g_pyr_lds) // both pyr_down kernels of the product are emulated (launch_pyr_down picks by image count)
                    launch((p.lw[l + 1] + PD_TW - 1) / PD_TW, (p.lh[l + 1] + PD_TH - 1) / PD_TH, n_img, 256, [&] { pyr_down_lds_kernel(d_imgs, l); });
                else
                    launch((p.lw[l + 1] + PN_TW - 1) / PN_TW, (p.lh[l + 1] + PN_TH - 1) / PN_TH, n_img, 256, [&] { pyr_down_kernel(d_imgs, l); });
        const BorderBlocks bb = border_blocks(first, last, p.ls, p.lh);
        launch(bb.first[last], n_img, 1, 256, [&] { border_fill_kernel(d_imgs, last, bb); });
        const ScharrTiles st = scharr_tiles(first, last, p.lw, p.lh);
        launch(st.first[last], n_img, 1, 256, [&] { scharr_kernel(d_imgs, last, st); });
    }
}

} // namespace

extern "C" {
void ke_set_pyr_lds(int on) { g_pyr_lds = on; }

// the pyramid pass's workgroup-id decode (multiply-high divisions, XCD-aware order) against plain division: every id of a launch
// over n images of a w x h level-0 shape (n = 0 or beyond it: the largest launch pass_images_per_launch allows, ids sampled: the first and last
// 2^20 and every 4099th between).  Returns 0, or 1 + the level at which an id decoded wrongly / the decode was not a bijection.
int ke_pass_decode_check(int w, int h, int max_level, int n)
{
    using namespace vo;
    int lw[VO_MAX_LEVELS], lh[VO_MAX_LEVELS], ls[VO_MAX_LEVELS], L = 0;
    for (int cw = w, ch = h;; L++) {
        lw[L] = cw; lh[L] = ch; ls[L] = (VO_BX + cw + VO_BY + 15) / 16 * 16;
        const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (L == max_level || L + 1 >= VO_MAX_LEVELS || nw <= 21 || nh <= 21) break;
        cw = nw; ch = nh;
    }
    L++;
    const PassPlan pp = pass_plan(L, lw, lh, ls);
    for (int l = 0; l < L; l++) {
        const uint32_t nci = (uint32_t)pp.nci[l], wpi = nci * (uint32_t)pp.gy[l];
        const uint32_t per = (uint32_t)pass_images_per_launch(pp, l), ni = n > 0 && (uint32_t)n < per ? (uint32_t)n : per; // (more images: a second launch)
        for (int remap = 0; remap < 2; remap++) {
            const uint32_t nwg = pass_grid(pp, l, (int)ni, remap);
            if ((uint64_t)nwg != (uint64_t)wpi * (remap ? (ni + 7) / 8 * 8 : ni))
                return 1 + l;
            uint64_t seen = 0, expect = 0; // order-independent checksum of the decoded linear positions
            const bool full = nwg <= (1u << 22);
            for (uint32_t id = 0; id < nwg; id = (full || id < (1u << 20) || id + (1u << 20) >= nwg) ? id + 1 : id + 4099) {
                uint32_t z, rem, by, bx;
                const bool live = pass_decode(pp, l, id, ni, remap, &z, &rem, &by, &bx);
                const uint32_t lin = remap ? id >> 3 : id;
                const uint32_t zt = remap ? (lin / wpi) * 8 + (id & 7) : lin / wpi, rt = lin % wpi;
                if (live != (zt < ni))
                    return 1 + l;
                if (!live)
                    continue;
                if (z != zt || rem != rt || by != rt / nci || bx != rt % nci || by >= (uint32_t)pp.gy[l] || bx >= nci)
                    return 1 + l;
                seen += (uint64_t)z * wpi + rem + 1;
            }
            if (full) {
                const uint64_t tot = (uint64_t)ni * wpi;
                expect = tot * (tot + 1) / 2;
                if (seen != expect)
                    return 1 + l;
            }
        }
    }
    return 0;
}


// essential.hip on the emulator, launch by launch as launch_essential does: findEssentialMat(RANSAC) + recoverPose of one frame.
// p0, p1: n x 2 pixels.  Returns EmResult::status; E, R: 9, t: 3, mask: n, dbg: {n_inliers, n_good, niters, best}
int ke_essential(const float *p0, const float *p1, int n, double focal, double ppx, double ppy, double prob, double threshold,
                 int max_iters, double *E, double *R, double *t, uint8_t *mask, int *dbg4)
{
    using namespace vo;
    const int cap = n > 8 ? n : 8, iters = max_iters;
    EmParams prm;
    prm.focal = focal;
    prm.ppx = ppx;
    prm.ppy = ppy;
    prm.prob = prob;
    prm.threshold = threshold;
    prm.max_iters = max_iters;
    std::vector<uint32_t> raw(RNG_TABLE);
    launch(1, 1, 1, 1, [&] { rng_table_kernel(raw.data(), RNG_TABLE); });
    std::vector<float2> a((size_t)cap), b((size_t)cap);
    for (int i = 0; i < n; i++) {
        a[i] = make_float2(p0[2 * i], p0[2 * i + 1]);
        b[i] = make_float2(p1[2 * i], p1[2 * i + 1]);
    }
    std::vector<double2> q0((size_t)cap), q1((size_t)cap);
    std::vector<int32_t> subsets((size_t)iters * 5);
    std::vector<double> models((size_t)EM_CHUNK * EM_MAX_MODELS * 9), bestE(9);
    std::vector<int> nmodels((size_t)EM_CHUNK), counts((size_t)EM_CHUNK * EM_MAX_MODELS);
    std::vector<uint8_t> m((size_t)cap);
    RansacState st;
    EmResult res;
    memset(&res, 0, sizeof(res));
    int n_pts = n;
    const double thr = prm.threshold / ((prm.focal + prm.focal) / 2);
    const float thr2 = (float)(thr * thr);
    launch((unsigned)(cap + 255) / 256, 1, 1, 256, [&] { em_normalise_kernel(a.data(), b.data(), 0, &n_pts, cap, prm, q0.data(), q1.data()); });
    const int n_chunks = (iters + EM_CHUNK - 1) / EM_CHUNK;
    for (int chunk = 0; chunk < n_chunks; chunk++) {
        launch(1, 1, 1, 64, [&] { ransac_subsets_kernel(&n_pts, 1, iters, chunk * EM_CHUNK, EM_CHUNK, raw.data(), RNG_TABLE, subsets.data(), &st); });
        launch(EM_CHUNK / 64, 1, 1, 64, [&] { em_solve_kernel<1>(q0.data(), q1.data(), &n_pts, cap, iters, chunk, subsets.data(), &st, models.data(), nmodels.data()); });
        launch(EM_CHUNK * EM_MAX_MODELS, 1, 1, 64, [&] { em_vote_kernel(q0.data(), q1.data(), &n_pts, cap, iters, chunk, thr2, &st, models.data(), nmodels.data(), counts.data()); });
        launch(1, 1, 1, 64, [&] { em_replay_kernel(&n_pts, 1, iters, prm.prob, chunk, models.data(), nmodels.data(), counts.data(), &st, bestE.data()); });
    }
    launch(1, 1, 1, 256, [&] { em_finish_kernel(q0.data(), q1.data(), &n_pts, cap, thr2, &st, bestE.data(), m.data(), &res); });
    memcpy(E, res.E, sizeof(res.E));
    memcpy(R, res.R, sizeof(res.R));
    memcpy(t, res.t, sizeof(res.t));
    memcpy(mask, m.data(), (size_t)n);
    dbg4[0] = res.n_inliers;
    dbg4[1] = res.n_good;
    dbg4[2] = res.niters;
    dbg4[3] = res.best;
    return res.status;
}

// post.hip on the emulator: deleteUnmatchFeaturesCircle (stage A) + checkValidMatch / removeInvalidPoints (stage B) of one
// frame's four tracking hops, then triangulation of the stage-B left / right points.
// pts: n x 2, trk: 4 x n x 2 (r0, r1, l1, l0_ret), status: 4 x n; outA: 5 x n x 2, outB: 4 x n x 2, xyz: n x 3
int ke_post(const float *pts, const float *trk, const uint8_t *status, int n, int threshold, const float *P_l, const float *P_r,
            float *outA, int32_t *idxA, int *nA, float *outB, int32_t *idxB, int *nB, float *xyz)
{
    const int cap = n > 1 ? n : 1;
    int n_pts = n;
    launch(1, 1, 1, 1024, [&] {
        vo::compact_kernel((const float2 *)pts, (const float2 *)trk, status, &n_pts, cap, threshold, (float2 *)outA, idxA, nA,
                           (float2 *)outB, idxB, nB);
    });
    launch((unsigned)(cap + 255) / 256, 1, 1, 256, [&] {
        vo::triangulate_kernel(P_l, P_r, (const float2 *)outB, (const float2 *)outB + cap, (size_t)4 * cap, nB, cap, xyz);
    });
    return 0;
}

// The whole pose solve of ONE frame on the CPU emulator, kernel by kernel in launch_pnp's order: raw RNG table -> per chunk
// (subsets -> EPnP -> votes -> control-flow replay; with split = 1 everything behind the first chunk in ransac_rest_kernel) -> the four-point frames' P3P / winner / inlier mask / Levenberg-Marquardt refinement.
// split = 1: the first chunk's EPnP as the four kernels small launches use (epnp_prepare / svd12_wave / epnp_approx /
// epnp_select), 0: epnp_kernel<1>.  first_chunk: 128 or 64 (big launches).  Returns the PnpResult fields and the inliers.
static int pnp_ransac_emulated(const float *xyz, const float *uv, int n, const float *K9, int iters, float reproj,
                               double confidence, int split, int first_chunk, double *rvec, double *tvec, int32_t *inliers,
                               int *n_inliers, int *dbg4, const vo::SeqTail &tail)
{
    using namespace vo;
    const int cap = n > 8 ? n : 8;
    PnpParams prm;
    prm.iters = iters;
    prm.reproj = reproj;
    prm.confidence = confidence;
    memcpy(prm.K, K9, sizeof(prm.K));
    std::vector<uint32_t> raw(RNG_TABLE);
    launch(1, 1, 1, 1, [&] { rng_table_kernel(raw.data(), RNG_TABLE); });
    std::vector<float> X(xyz, xyz + (size_t)3 * n);
    X.resize((size_t)3 * cap);
    std::vector<float2> U((size_t)cap);
    for (int i = 0; i < n; i++)
        U[i] = make_float2(uv[2 * i], uv[2 * i + 1]);
    std::vector<int32_t> subsets((size_t)iters * 5), inl((size_t)cap);
    std::vector<double> models((size_t)iters * 6), ws((size_t)VO_EPNP_WS_HYPS * VO_EPNP_WS_DOUBLES);
    std::vector<int> counts((size_t)iters);
    std::vector<double> lds((size_t)(144 + 12) * 64), gws((size_t)VO_EPNP_GWS_BLOCKS * VO_EPNP_UT_DOUBLES * 64);
    RansacState st;
    PnpResult res;
    memset(&res, 0, sizeof(res));
    int n_pts = n;
    emu::dyn_shared() = lds.data();
    for (int h0 = 0; h0 < iters;) {
        const int hn = h0 == 0 ? std::min(first_chunk, iters) : iters - h0;
        const unsigned eg = (unsigned)(hn + 63) / 64;
        if (split == 1 && h0 > 0) { // small launches: the rest of the solve in one launch (ransac_rest_kernel)
            std::vector<double> rest_ws((size_t)eg * VO_EPNP_UT_DOUBLES * 64); // (the test picks its own first chunk: sized by this launch)
            launch(eg, 1, 1, 64, [&] { ransac_rest_kernel<2>(X.data(), U.data(), 0, &n_pts, cap, subsets.data(), prm, &st, h0, hn, models.data(), counts.data(), raw.data(), RNG_TABLE, (int)eg, rest_ws.data()); });
            break;
        }
        launch(1, 1, 1, 64, [&] { ransac_subsets_kernel(&n_pts, 1, iters, h0, hn, raw.data(), RNG_TABLE, subsets.data(), &st); });
        if (split == 1 && h0 == 0) {
            launch(eg, 1, 1, 64, [&] { epnp_prepare_kernel(X.data(), U.data(), 0, &n_pts, cap, subsets.data(), prm, &st, h0, hn, ws.data()); });
            launch((unsigned)hn, 1, 1, 128, [&] { svd12_wave_kernel(&n_pts, prm, &st, h0, hn, ws.data()); });
            launch(eg, 1, 3, 64, [&] { epnp_approx_kernel(&n_pts, prm, &st, h0, hn, ws.data()); });
            launch(eg, 1, 1, 64, [&] { epnp_select_kernel(&n_pts, prm, &st, h0, hn, ws.data(), models.data()); });
        } else {
            // split == 2: the slim form's first chunk -- the 12 x 12 matrices in the global workspace instead of LDS
            if (split == 2 && h0 == 0 && eg <= (unsigned)VO_EPNP_GWS_BLOCKS)
                launch(eg, 1, 1, 64, [&] { epnp_kernel<1, true>(X.data(), U.data(), 0, &n_pts, cap, subsets.data(), prm, &st, h0, hn, models.data(), gws.data()); });
            else
                launch(eg, 1, 1, 64, [&] { epnp_kernel<1, false>(X.data(), U.data(), 0, &n_pts, cap, subsets.data(), prm, &st, h0, hn, models.data(), nullptr); });
        }
        launch((unsigned)hn, 1, 1, 64, [&] { vote_kernel(X.data(), U.data(), 0, &n_pts, cap, prm, models.data(), &st, h0, counts.data()); });
        launch(1, 1, 1, 64, [&] { ransac_replay_kernel(&n_pts, 1, prm, h0 + hn, counts.data(), &st); });
        h0 += hn;
    }
    launch(1, 1, 1, 256, [&] { select_refine_kernel<1>(X.data(), U.data(), 0, &n_pts, cap, prm, models.data(), &st, inl.data(), &res, tail); });
    emu::dyn_shared() = nullptr;
    memcpy(rvec, res.rvec, sizeof(res.rvec));
    memcpy(tvec, res.tvec, sizeof(res.tvec));
    *n_inliers = res.n_inliers;
    for (int i = 0; i < res.n_inliers && i < n; i++)
        inliers[i] = inl[i];
    dbg4[0] = res.niters;
    dbg4[1] = res.best_iter;
    dbg4[2] = res.max_good;
    dbg4[3] = res.lm_iters;
    return res.status;
}

int ke_pnp_ransac(const float *xyz, const float *uv, int n, const float *K9, int iters, float reproj, double confidence,
                  int split, int first_chunk, double *rvec, double *tvec, int32_t *inliers, int *n_inliers, int *dbg4)
{
    return pnp_ransac_emulated(xyz, uv, n, K9, iters, reproj, confidence, split, first_chunk, rvec, tvec, inliers, n_inliers,
                               dbg4, vo::SeqTail());
}

// the same with the lock-step loop's tail: thread 0 of the refinement kernel goes on with the Euler gates +
// integrateOdometryStereo + one trajectory row (vo_seqtail.h).  pose16: frame_pose in / out; rows27: the sequence's
// trajectory rows so far (n_rows of them) + room for one more; info8 of the new row out
int ke_pnp_ransac_tail(const float *xyz, const float *uv, int n, const float *K9, int split, double *pose16, double *rows27,
                       int *n_rows, int max_steps, int *info8, int *dbg4)
{
    std::vector<vo::SeqFrameInfo> info((size_t)max_steps);
    int active = 1;
    vo::SeqTail tail;
    tail.active = &active;
    tail.pose = pose16;
    tail.traj = rows27;
    tail.info = info.data();
    tail.n_rows = n_rows;
    tail.max_steps = max_steps;
    double rv[3], tv[3];
    std::vector<int32_t> inl((size_t)(n > 1 ? n : 1));
    int ninl = 0;
    const int row = *n_rows;
    const int rc = pnp_ransac_emulated(xyz, uv, n, K9, 500, 0.5f, (double)0.999f, split, 128, rv, tv, inl.data(), &ninl, dbg4, tail);
    if (row < max_steps)
        memcpy(info8, &info[row], sizeof(vo::SeqFrameInfo));
    return rc;
}

// currentVOFeatures of the sequence after a frame (seq_carry_kernel): stage-B pointsLeft_t1 + the erase-compacted ages
int ke_seq_carry(const float *outB, int nB, const int32_t *idxA, int nA, const int32_t *ages, int n_bucketed, int cap, int fcap,
                 float *feat, int32_t *fages, int *n_tracked, int *n_ages)
{
    int active = 1, overflow = 0, n_rows_carry = 0;
    std::vector<vo::SeqFrameInfo> info(1);
    launch(1, 1, 1, 256, [&] {
        vo::seq_carry_kernel(&active, (const float2 *)outB, &nB, idxA, &nA, ages, &n_bucketed, cap, fcap, (float2 *)feat, fages,
                             n_tracked, &overflow, &n_rows_carry, n_ages, info.data(), 1);
    });
    return 0;
}

// The four-kernel EPnP of pnp.hip on the CPU: epnp5_prepare (one lane) -> the 12 x 12 SVD by a 128-thread workgroup
// (emulated) + jacobi12_finish -> the three approximations taken separately -> epnp5_select, next to the one-piece solver
// epnp5_solve the one-kernel form runs.  xyz5: n x 15, uv5: n x 10 (f32), K: 3 x 3 f32; rt_split / rt_mono: n x 6 (rvec, tvec)
int ke_epnp_split(const float *xyz5, const float *uv5, const float *K, int n, double *rt_split, double *rt_mono)
{
    for (int q = 0; q < n; q++) {
        vo::Epnp5 e;
        std::vector<double> ut(144), w(12);
        int flag = 0;
        vo::epnp5_prepare<1>(xyz5 + 15 * q, uv5 + 10 * q, K, e, ut.data());
        launch(1, 1, 1, 128, [&] { vo::jacobi12_pipe_sweeps(ut.data(), w.data(), &flag, (int)threadIdx.x); });
        vo::jacobi12_finish(ut.data(), w.data());
        double L[60], rho[6], rep[3], R[3][9], t[3][3];
        for (int a = 0; a < 3; a++) { // (each approximation starts from the state the kernel loads from the workspace)
            vo::Epnp5 ea = e;
            vo::epnp5_L_rho<1>(ea, ut.data(), L, rho);
            rep[a] = a == 0   ? vo::epnp5_approx<1, 0>(ea, ut.data(), L, rho, R[0], t[0])
                     : a == 1 ? vo::epnp5_approx<1, 1>(ea, ut.data(), L, rho, R[1], t[1])
                              : vo::epnp5_approx<1, 2>(ea, ut.data(), L, rho, R[2], t[2]);
        }
        vo::epnp5_select(rep, R[0], R[1], R[2], t[0], t[1], t[2], rt_split + 6 * q, rt_split + 6 * q + 3);
        vo::epnp5_solve(xyz5 + 15 * q, uv5 + 10 * q, K, rt_mono + 6 * q, rt_mono + 6 * q + 3);
    }
    return 0;
}

// The Levenberg-Marquardt step's 6 x 6 solve as select_refine_kernel runs it -- wavefront sweeps with V, jacobi_finish and
// svd_backsubst on lane 0 -- next to solve_svd<6, 6>.  A: n x 36 row-major, b: n x 6; x_wave / x_serial: n x 6
int ke_solve6_wave(const double *A, const double *b, int n, double *x_wave, double *x_serial)
{
    std::vector<double> at((size_t)n * 36), vt((size_t)n * 36), w((size_t)n * 6);
    launch(n, 1, 1, 64, [&] {
        const int q = (int)blockIdx.x, lane = (int)threadIdx.x;
        double *At = at.data() + (size_t)q * 36, *Vt = vt.data() + (size_t)q * 36, *W = w.data() + (size_t)q * 6;
        if (lane == 0)
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 6; k++)
                    At[i * 6 + k] = A[(size_t)q * 36 + k * 6 + i];
        emu::barrier();
        vo::jacobi6v_wave_sweeps(At, W, Vt, lane);
        emu::barrier();
        if (lane == 0) {
            vo::jacobi_finish<6, true>(At, W, Vt);
            vo::svd_backsubst<6>(At, W, Vt, b + (size_t)q * 6, x_wave + (size_t)q * 6);
        }
    });
    for (int q = 0; q < n; q++)
        vo::solve_svd<6, 6>(A + (size_t)q * 36, b + (size_t)q * 6, x_serial + (size_t)q * 6);
    return 0;
}

// EPnP's 12 x 12 SVD: the wavefront-per-matrix sweeps of vo_svd_wide.h (four DPP rows = four independent pairs per step) +
// jacobi12_finish on lane 0, as svd12_wave_kernel runs them, next to the one-lane routine jacobi_svd<12, 12, false> the
// monolithic EPnP kernel uses.  mats: n x 144; wide / serial: n x 144 sorted, normalised rows
int ke_svd12_wide(const double *mats, int n, double *wide, double *serial)
{
    std::vector<double> w((size_t)n * 12);
    for (int q = 0; q < n; q++)
        memcpy(wide + (size_t)q * 144, mats + (size_t)q * 144, 144 * sizeof(double));
    std::vector<int> flags((size_t)n);
    launch(n, 1, 1, 128, [&] {
        const int q = (int)blockIdx.x, tid = (int)threadIdx.x;
        vo::jacobi12_pipe_sweeps(wide + (size_t)q * 144, w.data() + (size_t)q * 12, flags.data() + q, tid);
        if (tid == 0)
            vo::jacobi12_finish(wide + (size_t)q * 144, w.data() + (size_t)q * 12);
    });
    for (int q = 0; q < n; q++) {
        double d12[12];
        memcpy(serial + (size_t)q * 144, mats + (size_t)q * 144, 144 * sizeof(double));
        vo::jacobi_svd<12, 12, false, 1>(serial + (size_t)q * 144, d12, nullptr);
    }
    return 0;
}

// the whole bordered allocation of one level of one image after the emulated pyramid build: rows -VO_BY .. h + VO_BY - 1,
// `stride` bytes / dwords each starting at column -VO_BX (pixels poisoned with 0xA5 before the build, derivatives zero)
int ke_bordered_level(const uint8_t *img, int w, int h, int max_level, int level, uint8_t *pix_out, uint32_t *der_out,
                      int cap, int *lvl_w, int *lvl_h, int *lvl_stride)
{
    using namespace vo;
    Plan p = plan(w, h, max_level);
    if (level < 0 || level >= p.levels)
        return -1;
    Heap heap(p, 1, img, w, h);
    build_pyramids(p, heap.tab.data(), 1);
    const int n = (int)heap.level_elems(level);
    if (n > cap)
        return -2;
    memcpy(pix_out, heap.pix_block(0, level), n);
    memcpy(der_out, heap.der_block(0, level), 4 * (size_t)n);
    *lvl_w = p.lw[level];
    *lvl_h = p.lh[level];
    *lvl_stride = p.ls[level];
    return p.levels;
}

// Exhaustive check of the FAST pre-test: for every assignment of {similar, brighter, darker} to the 16 circle pixels
// (3^16 rings) the full TYPE_9_16 test implies the compass test.  Returns the number of violations; *n_corners = rings that
// are corners, *n_candidates = rings that pass the compass test.
long long ke_fast_compass_exhaustive(long long *n_corners, long long *n_candidates)
{
    using namespace vo;
    static const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    static const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    uint8_t patch[7 * 7];
    const int stride = 7, threshold = 20;
    long long bad = 0, corners = 0, cands = 0, total = 1;
    for (int k = 0; k < 16; k++)
        total *= 3;
    for (long long code = 0; code < total; code++) {
        memset(patch, 100, sizeof(patch));
        long long c = code;
        for (int k = 0; k < 16; k++, c /= 3) {
            const int s = (int)(c % 3);
            patch[(3 + dy[k]) * stride + 3 + dx[k]] = s == 0 ? 100 : s == 1 ? 121 : 79; // just beyond the threshold
        }
        const uint8_t *p = &patch[3 * stride + 3];
        const bool corner = fast_is_corner(p, stride, threshold), cand = fast_compass_candidate(p, stride, threshold);
        corners += corner;
        cands += cand;
        bad += corner && !cand;
    }
    *n_corners = corners;
    *n_candidates = cands;
    return bad;
}

// fast_compass_pair (two positions per packed instruction, what the tile kernel runs) against fast_compass_candidate (the
// scalar statement the exhaustive ring test above is about): every (centre, threshold) with the four compass pixels swept over
// the values around centre +- threshold and the ends of the byte range -- 9^4 combinations per (centre, threshold), the two
// lanes of the pair carrying different centres.  Returns the number of disagreements.
long long ke_fast_compass_pair_check()
{
    using namespace vo;
    long long bad = 0;
    uint8_t patch[2][7 * 7];
    const int stride = 7;
    static const int thresholds[] = {0, 1, 7, 20, 100, 200, 254, 255};
    for (int threshold : thresholds)
        for (int v0 = 0; v0 < 256; v0 += (v0 < 24 || v0 > 230 ? 1 : 5)) {
            const int v1 = (v0 * 7 + 13) & 255; // the other lane's centre
            int cand[2][9];
            for (int l = 0; l < 2; l++) {
                const int v = l ? v1 : v0;
                const int raw[9] = {0, 255, v, v + threshold, v + threshold + 1, v - threshold, v - threshold - 1, v + threshold - 1, v - threshold + 1};
                for (int k = 0; k < 9; k++)
                    cand[l][k] = raw[k] < 0 ? 0 : raw[k] > 255 ? 255 : raw[k];
            }
            for (int code = 0; code < 9 * 9 * 9 * 9; code++) {
                uint32_t c[4] = {0, 0, 0, 0};
                bool want[2];
                for (int l = 0; l < 2; l++) {
                    memset(patch[l], 0, sizeof(patch[l]));
                    const int k0 = code % 9, k4 = code / 9 % 9, k8 = code / 81 % 9, k12 = (code / 729 + 3 * l) % 9;
                    const int px[4] = {cand[l][k0], cand[l][k4], cand[l][k8], cand[l][k12]};
                    uint8_t *p = &patch[l][3 * stride + 3];
                    p[0] = (uint8_t)(l ? v1 : v0);
                    p[3 * stride] = (uint8_t)px[0];
                    p[3] = (uint8_t)px[1];
                    p[-3 * stride] = (uint8_t)px[2];
                    p[-3] = (uint8_t)px[3];
                    want[l] = fast_compass_candidate(p, stride, threshold);
                    for (int k = 0; k < 4; k++)
                        c[k] |= (uint32_t)px[k] << (16 * l);
                }
                const uint32_t r = fast_compass_pair((uint32_t)v0 | (uint32_t)v1 << 16, c[0], c[1], c[2], c[3], (uint32_t)threshold | (uint32_t)threshold << 16);
                bad += ((r & 0xffffu) != 0) != want[0];
                bad += ((r >> 16) != 0) != want[1];
            }
        }
    return bad;
}

void ke_set_lk_pair(int on) { g_lk_pair = on == 3 ? 2 : on; g_lk_split_fine = on == 3; } // 0: one launch, 1: pair kernel, 2: hops [0,1)+[1,4), 3: four launches
void ke_set_fast_big(int on) { g_fast_big = on; }
void ke_set_bucket_threads(int n) { g_bucket_threads = n; }

// imgs: n_img images of w x h (contiguous).  Builds every pyramid with the emulated kernels.
// lvl_out / der_out (optional): interior of level `want_level` of image 0 (w_l*h_l bytes / dwords).
// Then tracks pts [n][2] through the quad (0,1,2,3) with the emulated LK kernel.
int ke_run(const uint8_t *imgs, int n_img, int w, int h, int max_level, int want_level, uint8_t *lvl_out,
           uint32_t *der_out, int *lvl_w, int *lvl_h, const float *pts, int n, int max_count, double eps,
           float min_eig, int full_chain, float *trk /* [4][n][2] */, uint8_t *status /* [4][n] */)
{
    using namespace vo;
    Plan p = plan(w, h, max_level);
    Heap heap(p, n_img, imgs, w, h);
    std::vector<PyrImage> &tab = heap.tab;
    const PyrImage *d_imgs = tab.data();
    build_pyramids(p, d_imgs, n_img);

    if (want_level >= 0 && want_level < p.levels) {
        const int l = want_level;
        *lvl_w = p.lw[l];
        *lvl_h = p.lh[l];
        for (int y = 0; y < p.lh[l]; y++) {
            if (lvl_out)
                memcpy(lvl_out + (size_t)y * p.lw[l], tab[0].lvl[l] + (ptrdiff_t)y * p.ls[l], p.lw[l]);
            if (der_out)
                memcpy(der_out + (size_t)y * p.lw[l], tab[0].der[l] + (ptrdiff_t)y * p.ls[l], 4 * (size_t)p.lw[l]);
        }
    }
    if (n <= 0 || n_img < 4)
        return p.levels;

    Quad quad{0, 1, 2, 3};
    LkParams prm;
    prm.max_level = p.levels - 1;
    prm.max_count = max_count;
    prm.epsilon = eps * eps;
    prm.min_eig = min_eig;
    prm.full_chain = full_chain;
    std::vector<float2> out((size_t)4 * n);
    const int cap = n, n_frames = 1, fpg = 1, parts = 8, ppp = (n + parts - 1) / parts;
    if (g_lk_pair == 1) {
        const int ppp2 = ((n + 1) / 2 + parts - 1) / parts; // pairs per part
        for (unsigned b = 0; b < (unsigned)(8 * ppp2); b++) {
            emu::run_block(64, b, 0, 0, [&] {
                lk_circular_pair_kernel(d_imgs, &quad, (const float2 *)pts, &n, cap, n_frames, fpg, ppp2, out.data(), status, prm);
            });
        }
    } else if (g_lk_pair == 2) {
        // the synchronous calls' split chain (lk_hops_kernel): hop 0, then hops 1 .. 3 from what hop 0 left in trk / status;
        // ke_set_lk_pair(3): four launches of one hop each.  The outputs start as garbage the second launch must not rely on.
        memset(status, 0xA5, (size_t)4 * n);
        for (auto &o : out)
            o = make_float2(123456.f, -7.f);
        const int cuts[2][5] = {{0, 1, 4, 4, 4}, {0, 1, 2, 3, 4}};
        const int *cut = cuts[g_lk_split_fine ? 1 : 0];
        for (int k = 0; k < 4 && cut[k] < 4; k++)
            for (unsigned b = 0; b < (unsigned)(8 * ppp); b++) {
                emu::run_block(64, b, 0, 0, [&] {
                    lk_hops_kernel(d_imgs, &quad, (const float2 *)pts, &n, cap, n_frames, fpg, ppp, out.data(), status, prm, cut[k],
                                   cut[k + 1]);
                });
            }
    } else {
        for (unsigned b = 0; b < (unsigned)(8 * ppp); b++) {
            emu::run_block(64, b, 0, 0, [&] {
                lk_circular_kernel(d_imgs, &quad, (const float2 *)pts, &n, cap, n_frames, fpg, ppp, out.data(), status, prm);
            });
        }
    }
    memcpy(trk, out.data(), sizeof(float2) * 4 * (size_t)n);
    return p.levels;
}

// FAST + NMS + bucketing kernels of fast.hip on one image.  tracked / ages_in: the carried set
// (n_tracked points, n_ages ages, n_ages >= n_tracked).  bucket_size == 0: return the raw corners.
// the exact wave reductions of vo_dev.h on caller-chosen per-lane partials (64 each): out = {sum2 a, sum2 b,
// sum3 a, sum3 b, sum3 c} as f32
void ke_wave_sums(const int *a, const int *b, const int *c, float *out5)
{
    using namespace vo;
    launch(1, 1, 1, 64, [&] {
        const int l = threadIdx.x;
        float s0, s1, t0, t1, t2;
        wave_sum2_exact_f32(a[l], b[l], s0, s1);
        wave_sum3_exact_f32(a[l], b[l], c[l], t0, t1, t2);
        if (l == 0) {
            out5[0] = s0;
            out5[1] = s1;
            out5[2] = t0;
            out5[3] = t1;
            out5[4] = t2;
        }
    });
}

int ke_detect(const uint8_t *img, int w, int h, int threshold, int nonmax, int do_detect, const float *tracked,
              int n_tracked, const int *ages_in, int n_ages, int bucket_size, int fpb, float *out_pts, int *out_ages,
              int out_cap)
{
    using namespace vo;
    Plan p = plan(w, h, 0);
    Heap heap(p, 1, img, w, h); // (a one-level pyramid: nothing follows level 0)
    PyrImage &im = heap.tab[0];
    Quad quad{0, 0, 0, 0};
    const int fcap = 1 << 17;
    std::vector<int> rowcnt(h, 0), rowoff(h, -1), fages(fcap, 0);
    std::vector<float2> feat(fcap);
    memcpy(feat.data(), tracked, sizeof(float2) * n_tracked);
    memcpy(fages.data(), ages_in, sizeof(int) * n_ages);
    int n_new = -1, n_out = -1;
    const int segs = (w + 63) / 64;
    std::vector<unsigned long long> nmsmask((size_t)h * segs, 0xDEADBEEFDEADBEEFull);
    if (g_fast_big == 1)
        launch(segs, (h + 31) / 32, 1, 256, [&] { fast_tile_tall_kernel(&im, &quad, &do_detect, threshold, nonmax, nmsmask.data(), segs, rowcnt.data()); });
    else if (g_fast_big == 2)
        launch((segs + 1) / 2, (h + 31) / 32, 1, 256, [&] { fast_tile_big_kernel(&im, &quad, &do_detect, threshold, nonmax, nmsmask.data(), segs, rowcnt.data()); });
    else
        launch(segs, (h + 15) / 16, 1, 256, [&] { fast_tile_kernel(&im, &quad, &do_detect, threshold, nonmax, nmsmask.data(), segs, rowcnt.data()); });
    launch(1, 1, 1, 256, [&] { fast_rowscan_kernel(rowcnt.data(), rowoff.data(), h, &do_detect, &n_new); });
    launch((h + 4 * fast_nms_rows_per_wave(segs) - 1) / (4 * fast_nms_rows_per_wave(segs)), 1, 1, 256, [&] { fast_nms_write_kernel(nmsmask.data(), segs, h, &do_detect, rowoff.data(), &n_tracked, fcap, feat.data()); });
    if (bucket_size <= 0) {
        const int k = n_new < out_cap ? n_new : out_cap;
        memcpy(out_pts, feat.data() + n_tracked, sizeof(float2) * k);
        return n_new;
    }
    if (!bucket_grid_ok(w, h, bucket_size, fpb))
        return -2;
    const bool fine = (long long)(h / bucket_size + 1) * (w / bucket_size + 1) > BK_MAX_CELLS; // (launch_bucket's choice)
    launch(1, 1, 1, (unsigned)g_bucket_threads, [&] {
        if (fine)
            bucket_kernel<BK_FINE_CELLS>(feat.data(), fages.data(), &n_tracked, &n_new, fcap, h, w, bucket_size, fpb, (float2 *)out_pts,
                                         out_ages, &n_out, out_cap, nullptr, nullptr, nullptr);
        else
            bucket_kernel<BK_MAX_CELLS>(feat.data(), fages.data(), &n_tracked, &n_new, fcap, h, w, bucket_size, fpb, (float2 *)out_pts,
                                        out_ages, &n_out, out_cap, nullptr, nullptr, nullptr);
    });
    return n_out;
}

// seq_ingest_kernel (round 6: a persistent grid of single-wave workgroups that walk over the rows): n_pairs pairs of w x h
// images with a byte stride -> pitched destination images (pair i: images 2 i, 2 i + 1), n_waves workgroups
void ke_seq_ingest(const uint8_t *const *left, const uint8_t *const *right, int n_pairs, int w, int h, int stride, int pitch,
                   uint8_t *dst /* [2 n_pairs][h][pitch] */, int n_waves)
{
    using namespace vo;
    std::vector<SeqIngest> tab((size_t)n_pairs);
    for (int i = 0; i < n_pairs; i++) {
        tab[i].left = left[i];
        tab[i].right = right[i];
        tab[i].stride = stride;
        tab[i].image0 = 2 * i;
    }
    const int n_rows = 2 * n_pairs * h;
    launch((unsigned)n_waves, 1, 1, 64, [&] { seq_ingest_kernel(tab.data(), n_rows, n_waves, w, h, pitch, dst, (size_t)h * pitch); });
}
}
