// See dpm_solver.h.  The float/double mixture below is part of the contract: tables must be bit-identical
// to the reference's (value_type = float, dpm_solver.h:13; the grid step and the cumulative alpha product
// are carried in double, dpm_solver.cpp:15 and :91).  Verified bit-for-bit against the reference's own
// dpm_solver.cpp through tests/golden/dpm_steps{20,50}.json.
#include "dpm_solver.h"

#include <cmath>
#include <limits>
#include <stdexcept>

namespace sdod {
namespace {

// n-point grid from a to b; the increment is a double, the running value is rounded to float each step
std::vector<float> float_grid(float a, float b, unsigned n, unsigned skip) {
    std::vector<float> out;
    out.reserve(n - skip);
    const double inc = static_cast<double>(b - a) / static_cast<double>(n - 1);
    float v = a;
    for (unsigned i = 0; i < n; ++i) {
        if (i >= skip) out.push_back(v);
        v = static_cast<float>(static_cast<double>(v) + inc);
    }
    return out;
}

inline float lerp_through(float x, float x1, float y1, float x2, float y2) {
    const float slope = (y2 - y1) / (x2 - x1);
    return slope * (x - x1) + y1;
}

} // namespace

DpmSolver::DpmSolver(unsigned timesteps, float lin_start, float lin_end) : total_(timesteps) {
    if (timesteps < 2) throw std::invalid_argument("DpmSolver needs at least 2 training steps");
    all_t_ = float_grid(0.0f, 1.0f, timesteps + 1, 1);
    all_log_alpha_ = float_grid(std::sqrt(lin_start), std::sqrt(lin_end), timesteps, 0); // sqrt(beta_i)
    double cumulative = 1.0;
    for (float& v : all_log_alpha_) {
        const float alpha = 1 - v * v;
        cumulative *= alpha;
        v = static_cast<float>(0.5 * std::log(cumulative));
    }
}

const std::vector<float>& DpmSolver::prepare(unsigned steps) {
    if (steps < 1) throw std::invalid_argument("steps must be >= 1");
    const unsigned n = steps + 1;
    ts_ = float_grid(1.0f, static_cast<float>(1.0 / total_), n, 0);
    for (auto* v : {&log_alphas_, &lambdas_, &sigmas_, &alphas_, &phis_, &i2rs_, &model_ts_}) v->assign(n, 0.0f);

    // piecewise-linear lookup of log(alpha) on the training grid; ts_ is descending, so the cursor only moves down
    unsigned cursor = total_;
    const float inf = std::numeric_limits<float>::infinity();
    for (unsigned i = 0; i < n; ++i) {
        const float t = ts_[i];
        model_ts_[i] = static_cast<float>((static_cast<double>(t) - 1.0 / total_) * 1000);
        float la;
        if (t < all_t_.front() || t > all_t_.back()) {
            la = lerp_through(t, all_t_.back(), all_log_alpha_.back(), all_t_.front(), all_log_alpha_.front());
        } else {
            while (all_t_[cursor - 1] > t) --cursor;
            la = cursor >= total_ ? all_log_alpha_.back()
                                  : lerp_through(t, all_t_[cursor - 1], all_log_alpha_[cursor - 1], all_t_[cursor], all_log_alpha_[cursor]);
        }
        log_alphas_[i] = la;
        const float one_minus_a2 = 1 - std::exp(2 * la); // float exp
        lambdas_[i] = static_cast<float>(static_cast<double>(la) - 0.5 * static_cast<double>(std::log(one_minus_a2)));
        sigmas_[i] = std::sqrt(one_minus_a2);
        alphas_[i] = std::exp(la);
        phis_[i] = i >= 1 ? std::expm1(-(lambdas_[i] - lambdas_[i - 1])) : inf;
        if (i >= 2) {
            const float ratio = (lambdas_[i - 1] - lambdas_[i - 2]) / (lambdas_[i] - lambdas_[i - 1]);
            i2rs_[i] = static_cast<float>(1.0 / static_cast<double>(2 * ratio));
        } else {
            i2rs_[i] = inf;
        }
    }
    return model_ts_;
}

DpmSolver::StepCoef DpmSolver::coef(unsigned s) const {
    if (s + 1 >= ts_.size()) throw std::out_of_range("DpmSolver step out of range");
    StepCoef c{};
    c.order = s == 0 ? 1 : 2;
    c.sigma_s = sigmas_[s];
    c.alpha_s = alphas_[s];
    c.sigma_ratio = sigmas_[s + 1] / sigmas_[s];
    const float ap = alphas_[s + 1] * phis_[s + 1];
    if (c.order == 1) {
        c.c_prev = 0.0f;
        c.c_cur = -alphas_[s + 1] * phis_[s + 1];
    } else {
        c.c_prev = ap * i2rs_[s + 1];
        c.c_cur = -alphas_[s + 1] * phis_[s + 1] * (1 + i2rs_[s + 1]);
    }
    return c;
}

void DpmSolver::update_host(unsigned step, float* x, const float* eps, float* y_prev, unsigned n) const {
    const StepCoef c = coef(step);
    for (unsigned i = 0; i < n; ++i) {
        const float y = (x[i] + (-c.sigma_s) * eps[i]) / c.alpha_s;
        float xn = x[i] * c.sigma_ratio;
        if (c.order == 2) xn += c.c_prev * y_prev[i];
        xn += c.c_cur * y;
        x[i] = xn;
        y_prev[i] = y;
    }
}

const std::vector<float>& DpmSolver::table(int which) const {
    switch (which) {
    case 0: return ts_;
    case 1: return log_alphas_;
    case 2: return lambdas_;
    case 3: return sigmas_;
    case 4: return alphas_;
    case 5: return phis_;
    case 6: return i2rs_;
    case 7: return model_ts_;
    case 8: return all_t_;
    case 9: return all_log_alpha_;
    default: throw std::out_of_range("unknown DpmSolver table");
    }
}

} // namespace sdod
