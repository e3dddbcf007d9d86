// mppi/cost_base.hpp — the reference's CostBase (include/cost_base.hpp:7-176, src/cost_base.cpp)
// on top of the C-ABI: q(x) = (x-g)ᵀ diag(Q) (x-g), action cost λ uᵀ Σ⁻¹ ε, terminal = q(x_H).
#ifndef MPPI_COST_BASE_HPP_
#define MPPI_COST_BASE_HPP_

#include <vector>

#include "model_base.hpp"

class CostBase {
public:
    CostBase() = default;
    // cost_base.hpp:73-76 / cost_base.cpp:25-35; sigma [a*a], goal [s], Q [s] (the diagonal, Diag(in_Q))
    CostBase(const float lambda, const std::vector<float> &sigma, const std::vector<float> &goal, const std::vector<float> &Q)
        : m_lambda(lambda), m_sigma(sigma), m_goal(goal), m_Q(Q)
    {
        build();
    }
    CostBase(const float lambda, const std::vector<float> &sigma, const std::vector<float> &goal)
        : CostBase(lambda, sigma, goal, std::vector<float>(goal.size(), 1.f)) {}
    CostBase(const CostBase &o) : m_lambda(o.m_lambda), m_sigma(o.m_sigma), m_goal(o.m_goal), m_Q(o.m_Q) { if (o.m_h) build(); }
    CostBase &operator=(const CostBase &o)
    {
        if (this != &o) { CostBase t(o); std::swap(m_h, t.m_h); m_lambda = o.m_lambda; m_sigma = o.m_sigma; m_goal = o.m_goal; m_Q = o.m_Q; }
        return *this;
    }
    ~CostBase() { mppi_destroy(m_h); }

    // cost_base.hpp:98-100 — and, unlike the reference's graph constant, it takes effect
    bool setGoal(const std::vector<float> &goal)
    {
        if (!m_h || mppi_set_goal(m_h, goal.data(), (int)goal.size()) != MPPI_OK) return false;
        m_goal = goal;
        return true;
    }
    void setConsts() {} // Σ⁻¹ and Diag(Q) are built at construction (cost_base.cpp:37-41)

    std::vector<float> mStateCost(const std::vector<float> &state, int k)
    {
        std::vector<float> out(k);
        mppi_detail::check(mppi_state_cost(m_h, state.data(), k, out.data()), m_h, "mppi_state_cost");
        return out;
    }
    std::vector<float> mActionCost(const std::vector<float> &action, const std::vector<float> &noise, int k)
    {
        std::vector<float> out(k);
        mppi_detail::check(mppi_action_cost(m_h, action.data(), noise.data(), k, out.data()), m_h, "mppi_action_cost");
        return out;
    }
    std::vector<float> mBuildStepCostGraph(const std::vector<float> &state, const std::vector<float> &action,
                                           const std::vector<float> &noise, int k)
    {
        std::vector<float> out(k);
        mppi_detail::check(mppi_step_cost(m_h, state.data(), action.data(), noise.data(), k, out.data()), m_h, "mppi_step_cost");
        return out;
    }
    std::vector<float> mBuildFinalStepCostGraph(const std::vector<float> &state, int k) { return mStateCost(state, k); }

private:
    void build()
    {
        const int s = (int)m_goal.size();
        int a = 1;
        while (a * a < (int)m_sigma.size()) ++a;
        mppi_config cfg;
        mppi_detail::check(mppi_config_init(&cfg, 1, 1, 0.1f, 1.f, s, a), nullptr, "mppi_config_init");
        cfg.lambda = m_lambda;
        cfg.sigma = m_sigma.data();
        cfg.goal = m_goal.data();
        cfg.Q = m_Q.data();
        mppi_detail::check(mppi_create(&cfg, &m_h), nullptr, "mppi_create");
    }
    float m_lambda = 1.f;
    std::vector<float> m_sigma, m_goal, m_Q;
    mppi_handle *m_h = nullptr;
};

#endif
