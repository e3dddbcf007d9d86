// mppi/model_base.hpp — the reference's ModelBase (include/model_base.hpp:13-134, src/model_base.cpp)
// on top of the C-ABI. Same constructor and method names; where the reference builds TF graph nodes
// from (Scope, Input) the methods here evaluate eagerly on the GPU from host vectors
// (state [kx*s] row-major with kx in {1,k}; action [k*a]) and return host vectors.
#ifndef MPPI_MODEL_BASE_HPP_
#define MPPI_MODEL_BASE_HPP_

#include <stdexcept>
#include <string>
#include <vector>

#include "../mppi_c.h"

namespace mppi_detail {
inline void check(mppi_status st, const mppi_handle *h, const char *what)
{
    if (st != MPPI_OK)
        throw std::runtime_error(std::string(what) + ": " + mppi_status_string(st) + " — " + mppi_last_error(h));
}
} // namespace mppi_detail

class ModelBase {
public:
    ModelBase() : ModelBase(1.f, 0.01f, 2, 1) {} // model_base.cpp:13-15 defaults
    ModelBase(const float mass, const float dt, const int s_dim, const int a_dim)
        : m_m(mass), m_dt(dt), m_s_dim(s_dim), m_a_dim(a_dim)
    {
        mppi_config cfg;
        mppi_detail::check(mppi_config_init(&cfg, 1, 1, dt, mass, s_dim, a_dim), nullptr, "mppi_config_init");
        mppi_detail::check(mppi_create(&cfg, &m_h), nullptr, "mppi_create");
    }
    ModelBase(const ModelBase &o) : ModelBase(o.m_m, o.m_dt, o.m_s_dim, o.m_a_dim) {}
    ModelBase &operator=(const ModelBase &o)
    {
        if (this != &o) { ModelBase t(o); std::swap(m_h, t.m_h); m_m = o.m_m; m_dt = o.m_dt; m_s_dim = o.m_s_dim; m_a_dim = o.m_a_dim; }
        return *this;
    }
    ~ModelBase() { mppi_destroy(m_h); }

    // A x  (model_base.cpp:59-68); returns [kx*s]
    std::vector<float> mBuildFreeStepGraph(const std::vector<float> &state, int kx)
    {
        std::vector<float> v((size_t)kx * m_a_dim, 0.f), out((size_t)kx * m_s_dim);
        mppi_detail::check(mppi_model_step(m_h, state.data(), kx, v.data(), kx, out.data(), nullptr, nullptr), m_h, "mppi_model_step");
        return out;
    }
    // (B/m) u  (model_base.cpp:70-82); returns [k*s]
    std::vector<float> mBuildActionStepGraph(const std::vector<float> &action, int k)
    {
        std::vector<float> x((size_t)k * m_s_dim, 0.f), out((size_t)k * m_s_dim);
        mppi_detail::check(mppi_model_step(m_h, x.data(), k, action.data(), k, nullptr, out.data(), nullptr), m_h, "mppi_model_step");
        return out;
    }
    // A x + (B/m) u, broadcasting a single state row over k actions (model_base.cpp:53-57); returns [k*s]
    std::vector<float> mBuildModelStepGraph(const std::vector<float> &state, int kx, const std::vector<float> &action, int k)
    {
        std::vector<float> out((size_t)k * m_s_dim);
        mppi_detail::check(mppi_model_step(m_h, state.data(), kx, action.data(), k, nullptr, nullptr, out.data()), m_h, "mppi_model_step");
        return out;
    }
    float mass() const { return m_m; }
    float dt() const { return m_dt; }

private:
    float m_m, m_dt;
    int m_s_dim, m_a_dim;
    mppi_handle *m_h = nullptr;
};

#endif
