// C++ parity test of include/mppi/*.hpp against the reference's own gtest vectors
// (test/test_model.cpp:120-255, test/test_cost.cpp:169-239, test/test_controller.cpp:71-222),
// written to read like them. gtest is not in the image: EXPECT_FLOAT_EQ is re-created (4 ulp).
// Needs a GPU (every numeric call runs HIP kernels); run by tests/test_parity_gpu.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mppi/controller_base.hpp"

using namespace std;
static int g_fail = 0;

static int64_t key(float x) { int32_t i; memcpy(&i, &x, 4); return i < 0 ? -(int64_t)(i & 0x7FFFFFFF) : i; }
static void expect_float_eq(float got, float exp, const string &what, int idx)
{
    if (llabs(key(got) - key(exp)) > 4) { printf("FAIL %s[%d]: got %.9g expected %.9g\n", what.c_str(), idx, got, exp); ++g_fail; }
}
static void test_tensor(const vector<float> &computed, const vector<float> &expected, const string &name)
{
    if (computed.size() != expected.size()) { printf("FAIL %s: size %zu vs %zu\n", name.c_str(), computed.size(), expected.size()); ++g_fail; return; }
    for (size_t i = 0; i < expected.size(); i++) expect_float_eq(computed[i], expected[i], name, (int)i);
}

static void ModelBaseTest_LargeTesting_and_InitTest()
{
    const int k3 = 5, s_dim3 = 6, a_dim3 = 3;
    const float m3 = 1.5f, dt3 = 0.01f;
    vector<float> state3 = {0., 0., 0., 0., 0., 0., 2., 1., 5., 0., -1., -2., 0.5, 0.5, 0.5, 0.5, 0.5, 0.5,
                            1., 0., 1., 0., 1., 0., -1, 0.5, -3, 2., 0., 0.};
    vector<float> state_init = {-1, 0.5, -3, 2., 0., 0.};
    vector<float> action3 = {1., 1., 1., 2., 0., -1., 0., 0., 0., 0.5, -0.5, 0.5, 3., 3., 3.};
    ModelBase model3(m3, dt3, s_dim3, a_dim3);
    float acc = (dt3 * dt3) / (2.f * m3);
    float vel = (dt3) / (m3);
    vector<float> exp_u = {acc, vel, acc, vel, acc, vel, 2.f * acc, 2.f * vel, 0.f * acc, 0 * vel, -1.f * acc, -1.f * vel,
                           0.f * acc, 0.f * vel, 0.f * acc, 0 * vel, 0.f * acc, 0.f * vel,
                           0.5f * acc, 0.5f * vel, -0.5f * acc, -0.5f * vel, 0.5f * acc, 0.5f * vel,
                           3.f * acc, 3.f * vel, 3.f * acc, 3.f * vel, 3.f * acc, 3.f * vel};
    vector<float> exp_s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 2.f + dt3, 1.f, 5.f, 0.f, -1.f - 2.f * dt3, -2.f,
                           0.5f + dt3 / 2.f, 0.5f, 0.5f + dt3 / 2.f, 0.5f, 0.5f + dt3 / 2.f, 0.5f,
                           1.f, 0.f, 1.f, 0.f, 1.f, 0.f, -1.f + dt3 / 2.f, 0.5f, -3.f + 2.f * dt3, 2.f, 0.f, 0.f};
    vector<float> exp_res;
    for (int i = 0; i < k3 * s_dim3; i++) exp_res.push_back(exp_u[i] + exp_s[i]);
    test_tensor(model3.mBuildFreeStepGraph(state3, k3), exp_s, "LargeTesting free");
    test_tensor(model3.mBuildActionStepGraph(action3, k3), exp_u, "LargeTesting action");
    test_tensor(model3.mBuildModelStepGraph(state3, k3, action3, k3), exp_res, "LargeTesting result");
    // InitTest: one state row broadcast over k actions (test_model.cpp:216-255)
    vector<float> exp_s1(exp_s.end() - 6, exp_s.end()), exp_res1;
    for (int i = 0; i < k3 * s_dim3; i++) exp_res1.push_back(exp_u[i] + exp_s1[i % 6]);
    test_tensor(model3.mBuildFreeStepGraph(state_init, 1), exp_s1, "InitTest free");
    test_tensor(model3.mBuildModelStepGraph(state_init, 1, action3, k3), exp_res1, "InitTest result");
}

static void CostBaseTest_StateCost_StepCost()
{
    const int k3 = 5;
    vector<float> state3 = {0., 0.5, 2., 0., 0., 2., 0., 0., 10., 2., 2., 3, 1., 1., 1., 2., 3., 4., 5., 6.};
    vector<float> goal3 = {1., 1., 1., 2.};
    vector<float> action3 = {0.5, 2., 0.25};
    vector<float> epsilon3 = {0.5, 1., 2., 0.5, 2., 0.25, -2, -0.2, -1, 0, 0, 0, 1., 0.5, 3.};
    vector<float> sig3 = {1., 0., 0., 0., 1., 0., 0., 0., 1.};
    vector<float> q3 = {1., 1., 10., 10.};
    CostBase c3(1., sig3, goal3, q3);
    c3.setConsts();
    test_tensor(c3.mBuildFinalStepCostGraph(state3, k3), {51.25, 52, 102, 0., 333}, "StateCost");
    test_tensor(c3.mBuildStepCostGraph(state3, action3, epsilon3, k3),
                {51.25 + 2.75, 52 + 4.3125, 102 - 1.65, 0. + 0, 333 + 2.25}, "StepCost");
}

static void ControllerBaseTest_testUpdate_testNew_testShift()
{
    const int k = 5, tau = 3, a_dim = 2;
    ControllerBase cont(k, tau, 0.01, 1., 4, a_dim);
    vector<float> cost = {3., 10., 0., 1., 5.};
    vector<float> noise = {1., -0.5, 1., -0.5, 2., 1., 0.3, 0, 2., 0.2, 1.2, 3., 0.5, 0.5, 0.5, 0.5, 0.5, 0.5,
                           0.6, 0.7, 0.2, -0.3, 0.1, -0.4, -2., -3., -4., -1., 0., 0.};
    vector<float> action = {1., 0.5, 2.3, 4.5, 2.1, -0.4};
    test_tensor(cont.mPrepareAction(action, 1), {2.3, 4.5}, "a1");
    test_tensor(cont.mPrepareNoise(noise, 2), {2., 1., 1.2, 3., 0.5, 0.5, 0.1, -0.4, 0., 0.}, "n2");
    auto t = cont.mBuildUpdateGraph(cost, noise, action);
    vector<float> w = {0.034951787275480706, 3.1871904480408675e-05, 0.7020254138530686, 0.2582607169364174, 0.004730210030553017};
    expect_float_eq(t.beta, 0.f, "beta", 0);
    test_tensor(t.exp_arg, {-3., -10., 0, -1., -5.}, "exp_arg");
    test_tensor(t.exp, {0.049787068367863944, 4.5399929762484854e-05, 1, 0.36787944117144233, 0.006737946999085467}, "exp");
    expect_float_eq(t.nabla, 1.424449856468154f, "nabla", 0);
    test_tensor(t.weights, w, "weights");
    vector<float> expected;
    for (int c = 0; c < tau * a_dim; c++) {
        double s = 0;
        for (int i = 0; i < k; i++) s += (double)w[i] * noise[i * tau * a_dim + c];
        expected.push_back((float)s);
    }
    test_tensor(t.weighted_noise, expected, "weighted noise");
    float sum_w = 0;
    for (float x : t.weights) sum_w += x;
    expect_float_eq(sum_w, 1.f, "sum_w", 0);
    test_tensor(cont.mGetNew(action, 2), {1, 0.5, 2.3, 4.5}, "testNew 2");
    test_tensor(cont.mShift(action, {1, 0.5}, 1), {2.3, 4.5, 2.1, -0.4, 1., 0.5}, "testShiftAndInit 1");
    test_tensor(cont.mShift(action, {1, 0.5, 2.3, 4.5}, 2), {2.1, -0.4, 1., 0.5, 2.3, 4.5}, "testShiftAndInit 2");
    if (cont.setGoal({1.f, 2.f, 3.f})) { printf("FAIL setGoal accepted a wrong size\n"); ++g_fail; }
}

int main()
{
    try {
        ModelBaseTest_LargeTesting_and_InitTest();
        CostBaseTest_StateCost_StepCost();
        ControllerBaseTest_testUpdate_testNew_testShift();
    } catch (const std::exception &e) {
        printf("FAIL exception: %s\n", e.what());
        return 2;
    }
    printf(g_fail ? "%d FAILED\n" : "all reference vectors pass (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
