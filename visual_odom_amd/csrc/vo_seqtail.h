// vo_seqtail.h -- tail of the reference's frame loop for ONE sequence of the lock-step loop, on the device:
// rotationMatrixToEulerAngles + the |euler| < 0.1 gate + integrateOdometryStereo (main.cpp:196-208, utils.cpp:57-131;
// vo_integrate.h, the same code vo_integrate_odometry() runs on the host) and one trajectory row.  Called by thread 0 of
// the sequence's select_refine_kernel workgroup (pnp.hip) right after it has written the frame's PnpResult.
#pragma once

#include "vo_integrate.h"
#include "vo_kernels.h"

namespace vo {

__device__ inline void seq_integrate_frame(const SeqTail &s, int f, const PnpResult &r, int active)
{
    double *P = s.pose + (size_t)f * 16;
    int flags = VO_SEQ_F_ACTIVE | ((active & 2) ? VO_SEQ_F_GAP : 0);
    float euler[3] = {0.f, 0.f, 0.f};
    double R[9], t[3] = {0, 0, 0}, rv[3] = {0, 0, 0};
    for (int k = 0; k < 9; k++)
        R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    const int row = s.n_rows[f];
    if (r.status < 0) {
        flags |= VO_SEQ_F_TOO_FEW; // the reference's solvePnPRansac asserts here; the pose stays
    } else {
        const bool untouched = r.status == 0 && r.lm_iters < 0;
        for (int k = 0; k < 3; k++) {
            t[k] = r.tvec[k];
            rv[k] = r.rvec[k];
        }
        if (untouched) {
            // four points and P3P found no solution: solvePnP returned false with the shared buffers UNTOUCHED --
            // rvec is the zeros of visualOdometry.cpp:162, `translation` is still the previous frame's (main.cpp:82)
            const bool have_prev = row > 0 && row <= s.max_steps;
            for (int k = 0; k < 3; k++) {
                rv[k] = 0;
                t[k] = have_prev ? s.traj[((size_t)f * s.max_steps + row - 1) * VO_SEQ_ROW + 15 + k] : 0.0;
            }
        }
        bool have_R = true;
        if (s.em) { // mono_rotation: rotation = recoverPose's (visualOdometry.cpp:146-157)
            const EmResult e = s.em[f];
            if (e.status == 1) {
                for (int k = 0; k < 9; k++)
                    R[k] = e.R[k];
            } else {
                have_R = false;
                flags |= VO_SEQ_F_NO_ESSENTIAL; // recoverPose throws on the empty E in the reference
            }
        } else if (!untouched) { // (untouched: R stays Rodrigues(0) = identity)
            for (int k = 0; k < 9; k++)
                R[k] = r.R[k];
        }
        if (have_R && integrate_odometry(P, R, t, euler))
            flags |= VO_SEQ_F_INTEGRATED;
    }
    if (row < s.max_steps) {
        double *o = s.traj + ((size_t)f * s.max_steps + row) * VO_SEQ_ROW;
        for (int k = 0; k < 12; k++)
            o[k] = P[k];
        for (int k = 0; k < 3; k++) {
            o[12 + k] = rv[k];
            o[15 + k] = t[k];
        }
        for (int k = 0; k < 9; k++)
            o[18 + k] = R[k];
        SeqFrameInfo &q = s.info[(size_t)f * s.max_steps + row];
        q.n_inliers = r.status >= 0 ? r.n_inliers : 0;
        q.pnp_status = r.status;
        q.flags = flags;
        q.ransac_iters = r.niters;
    }
    s.n_rows[f] = row + 1;
}

} // namespace vo
