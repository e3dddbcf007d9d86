// mppi/controller_base.hpp — the reference's ControllerBase (include/controller_base.hpp:14-376,
// src/controller_base.cpp) as a thin C++ class over the C-ABI (include/mppi_c.h): same constructor,
// next / setGoal / saveNext / toCSV, and the public helpers the reference's tests call.
// A host loop written against the reference (src/main.cpp:30-45) compiles against this header
// by changing the include path and linking libmppi_hip.so instead of TensorFlow.
#ifndef MPPI_CONTROLLER_BASE_HPP_
#define MPPI_CONTROLLER_BASE_HPP_

#include <iostream>
#include <string>
#include <vector>

#include "cost_base.hpp"
#include "model_base.hpp"

class ControllerBase {
public:
    // controller_base.cpp:23-71: λ=1, Σ=I, Q=1, goal=(1,0) per axis; the model is built with the
    // literal mass 1. (:68), not `mass` — reproduced; pass a config through the second ctor to differ.
    ControllerBase(const int k, const int tau, const float dt, const float mass, const int s_dim, const int a_dim)
        : m_k(k), m_tau(tau), m_s_dim(s_dim), m_a_dim(a_dim), m_dt(dt), m_mass(mass)
    {
        mppi_config cfg;
        mppi_detail::check(mppi_config_init(&cfg, k, tau, dt, 1.f, s_dim, a_dim), nullptr, "mppi_config_init");
        mppi_detail::check(mppi_create(&cfg, &m_h), nullptr, "mppi_create");
        // the reference's next() appends (x, u) to m_db on every call (controller_base.cpp:146-147): the log is on,
        // as a ring of kLogRows transitions (the reference's vectors grow without bound)
        mppi_detail::check(mppi_set_transition_log(m_h, kLogRows), m_h, "mppi_set_transition_log");
    }
    static constexpr int kLogRows = 1 << 16;
    // everything the reference hard-codes, as the caller's mppi_config (K-sharding, Σ, goal, seed, ...)
    explicit ControllerBase(const mppi_config &cfg)
        : m_k(cfg.k), m_tau(cfg.tau), m_s_dim(cfg.s_dim), m_a_dim(cfg.a_dim), m_dt(cfg.dt), m_mass(cfg.mass)
    {
        mppi_detail::check(mppi_create(&cfg, &m_h), nullptr, "mppi_create");
        mppi_detail::check(mppi_set_transition_log(m_h, kLogRows), m_h, "mppi_set_transition_log");
    }
    ControllerBase(const ControllerBase &) = delete;
    ControllerBase &operator=(const ControllerBase &) = delete;
    ~ControllerBase() { mppi_destroy(m_h); }

    // controller_base.cpp:126-133
    bool setGoal(std::vector<float> goal)
    {
        if ((int)goal.size() != m_s_dim) {
            std::cerr << "Wrong goal size, it should match the state dimension: " << m_s_dim << std::endl;
            return false;
        }
        return mppi_set_goal(m_h, goal.data(), (int)goal.size()) == MPPI_OK;
    }
    // controller_base.cpp:135-153
    std::vector<float> next(std::vector<float> x)
    {
        std::vector<float> act(m_a_dim);
        mppi_detail::check(mppi_next(m_h, x.data(), (int)x.size(), act.data(), m_a_dim), m_h, "mppi_next");
        return act;
    }
    // the same step with injected noise [k*tau*a] (what the reference's tests do by hand)
    std::vector<float> nextWithNoise(const std::vector<float> &x, const std::vector<float> &noise)
    {
        std::vector<float> act(m_a_dim);
        mppi_detail::check(mppi_next_with_noise(m_h, x.data(), (int)x.size(), noise.data(), noise.size(), act.data(), m_a_dim), m_h, "mppi_next_with_noise");
        return act;
    }
    void saveNext(std::vector<float> x_next) { mppi_detail::check(mppi_save_next(m_h, x_next.data(), (int)x_next.size()), m_h, "mppi_save_next"); }
    void toCSV(std::string filename) { mppi_detail::check(mppi_to_csv(m_h, filename.c_str()), m_h, "mppi_to_csv"); }
    // Options of the Python reference's update, not in the C++ one: clip_act (controller_base.py:500-504) and the
    // Savitzky-Golay filterSeq (controller_base.py:277-291). Empty vectors / window 0 switch them off.
    void setActionLimits(const std::vector<float> &a_min, const std::vector<float> &a_max)
    {
        mppi_detail::check(mppi_set_action_limits(m_h, a_min.empty() ? nullptr : a_min.data(), a_max.empty() ? nullptr : a_max.data(),
                                                  (int)a_min.size()), m_h, "mppi_set_action_limits");
    }
    void setSequenceFilter(int window, int polyorder) { mppi_detail::check(mppi_set_sequence_filter(m_h, window, polyorder), m_h, "mppi_set_sequence_filter"); }

    // mBeta … mWeightedNoise + mBuildUpdateGraph (controller_base.cpp:166-192, 215-224), fused on the device
    struct UpdateTerms {
        float beta = 0, nabla = 0;
        std::vector<float> exp_arg, exp, weights, weighted_noise, update;
    };
    UpdateTerms mBuildUpdateGraph(const std::vector<float> &cost, const std::vector<float> &noises, const std::vector<float> &actions)
    {
        UpdateTerms t;
        t.exp_arg.resize(m_k); t.exp.resize(m_k); t.weights.resize(m_k);
        t.weighted_noise.resize((size_t)m_tau * m_a_dim); t.update.resize((size_t)m_tau * m_a_dim);
        mppi_detail::check(mppi_update(m_h, cost.data(), noises.data(), actions.data(), &t.beta, t.exp_arg.data(), t.exp.data(),
                                       &t.nabla, t.weights.data(), t.weighted_noise.data(), t.update.data()), m_h, "mppi_update");
        return t;
    }
    // mBuildModelGraph (controller_base.cpp:226-273): rollout costs [k]
    std::vector<float> mBuildModelGraph(const std::vector<float> &init_state, const std::vector<float> &actions, const std::vector<float> &noises)
    {
        std::vector<float> c(m_k);
        mppi_detail::check(mppi_rollout_cost(m_h, init_state.data(), actions.data(), noises.data(), c.data()), m_h, "mppi_rollout_cost");
        return c;
    }
    // mPrepareAction / mPrepareNoise (controller_base.cpp:205-213): slices of the [tau,a] / [k,tau,a] layouts
    std::vector<float> mPrepareAction(const std::vector<float> &actions, int timestep) const
    {
        return std::vector<float>(actions.begin() + (size_t)timestep * m_a_dim, actions.begin() + (size_t)(timestep + 1) * m_a_dim);
    }
    std::vector<float> mPrepareNoise(const std::vector<float> &noises, int timestep) const
    {
        std::vector<float> out((size_t)m_k * m_a_dim);
        for (int k = 0; k < m_k; ++k)
            for (int j = 0; j < m_a_dim; ++j) out[(size_t)k * m_a_dim + j] = noises[((size_t)k * m_tau + timestep) * m_a_dim + j];
        return out;
    }
    // mGetNew / mInit0 / mShift (controller_base.cpp:310-329)
    std::vector<float> mGetNew(const std::vector<float> &current, int nb) const
    {
        std::vector<float> out((size_t)nb * m_a_dim);
        mppi_get_new(current.data(), m_tau, m_a_dim, nb, out.data());
        return out;
    }
    std::vector<float> mInit0(int nb) const { return std::vector<float>((size_t)nb * m_a_dim, 0.f); }
    std::vector<float> mShift(const std::vector<float> &current, const std::vector<float> &init, int nb) const
    {
        const int nb_init = (int)(init.size() / m_a_dim);
        std::vector<float> out((size_t)(m_tau - nb + nb_init) * m_a_dim);
        mppi_shift(current.data(), m_tau, m_a_dim, init.data(), nb_init, nb, out.data());
        return out;
    }
    std::vector<float> actionSequence()
    {
        std::vector<float> U((size_t)m_tau * m_a_dim);
        mppi_detail::check(mppi_get_action_sequence(m_h, U.data(), (int)U.size()), m_h, "mppi_get_action_sequence");
        return U;
    }
    mppi_handle *handle() { return m_h; }

private:
    int m_k, m_tau, m_s_dim, m_a_dim;
    float m_dt, m_mass;
    mppi_handle *m_h = nullptr;
};

#endif
