// DPM-Solver++(2M) host tables for the SD "scaled-linear" VP schedule.
// Same public behaviour as the reference's libsdod::DPMSolver (csrc/libsdod/src/dpm_solver.h:11-50,
// dpm_solver.cpp:84-181): tables over the training steps, per-run tables for `steps` ODE steps, and the
// coefficients of the multistep update.  The update itself runs on the GPU (sdod_dpm_update); the host
// version below exists for the C-ABI helper used by tests and for small vectors.
#pragma once
#include <vector>

namespace sdod {

class DpmSolver {
public:
    DpmSolver(unsigned timesteps, float lin_start, float lin_end);

    // fills the per-run tables; returns the model times (length steps + 1)
    const std::vector<float>& prepare(unsigned steps);

    struct StepCoef {
        int order;         // 1 at step 0, 2 afterwards (reference quirk Q8, dpm_solver.cpp:137)
        float sigma_s;     // x0 = (x - sigma_s * eps) / alpha_s
        float alpha_s;
        float sigma_ratio; // sigma[s+1] / sigma[s]
        float c_prev;      // alpha[s+1]*phi[s+1]*i2r[s+1]            (order 2 only)
        float c_cur;       // -alpha[s+1]*phi[s+1]  or  -alpha[s+1]*phi[s+1]*(1 + i2r[s+1])
    };
    StepCoef coef(unsigned step) const;

    // host reference of one update on small vectors (same arithmetic order as the device kernel)
    void update_host(unsigned step, float* x, const float* eps, float* y_prev, unsigned n) const;

    unsigned steps() const { return ts_.empty() ? 0 : (unsigned)ts_.size() - 1; }
    const std::vector<float>& table(int which) const; // 0 ts,1 log_alphas,2 lambdas,3 sigmas,4 alphas,5 phis,6 i2rs,7 model_ts,8 all_t,9 all_log_alpha

private:
    unsigned total_;
    std::vector<float> all_t_, all_log_alpha_;
    std::vector<float> ts_, log_alphas_, lambdas_, sigmas_, alphas_, phis_, i2rs_, model_ts_;
};

} // namespace sdod
