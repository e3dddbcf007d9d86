// examples/host_loop.cpp — the reference's host loop (src/main.cpp:16-67) against the MI355X library.
// The reference's loop is: get_x -> ctrl.next(x) -> env.simulate(u) -> get_x -> ctrl.saveNext(x);
// then ctrl.toCSV(...), then a (commented-out) chrono loop around ctrl.next. MuJoCo is replaced by the
// same point-mass plant stepped on the host (x' = A x + B u, envs/point_mass*.xml dimensions).
//   usage: host_loop [k=65536] [tau=64] [a_dim=3] [steps=100] [csv | -] [trace | armed=N]
// "armed=N" as the 6th argument: armed launches with a soft deadline of N microseconds (MPPI_TUNE_ARMED_US) — the next step's launch
// sits on the GPU, its noise drawn, while this loop steps the plant; next(x) then only stores x and waits for u (include/mppi_c.h).
// "trace" as the 6th argument switches on the library's roctx ranges (mppi:step > mppi:rollout / mppi:finish; the reference brackets its
// step with tf.profiler, controller_base.py:241-248): `rocprofv3 --marker-trace --kernel-trace -- examples/host_loop 65536 64 3 20 - trace`.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mppi/controller_base.hpp"

using namespace std;

int main(int argc, char const *argv[])
{
    int k = argc > 1 ? atoi(argv[1]) : 65536, tau = argc > 2 ? atoi(argv[2]) : 64;
    int aDim = argc > 3 ? atoi(argv[3]) : 3, steps = argc > 4 ? atoi(argv[4]) : 100;
    int sDim = 2 * aDim;
    float dt(0.1);
    try {
        ControllerBase ctrl(k, tau, dt, 1., sDim, aDim);
        vector<float> goal;
        const float pos[3] = {1.f, 0.5f, 0.75f}; // target site of envs/point_mass3d.xml:35
        for (int i = 0; i < aDim; i++) { goal.push_back(pos[i % 3]); goal.push_back(0.f); }
        if (!ctrl.setGoal(goal)) return 2;
        if (argc > 6 && std::string(argv[6]) == "trace" && mppi_set_tuning(ctrl.handle(), MPPI_TUNE_TRACE, 1) != MPPI_OK) {
            fprintf(stderr, "host_loop: %s\n", mppi_last_error(ctrl.handle()));
            return 4;
        }
        if (argc > 6 && std::string(argv[6]).rfind("armed=", 0) == 0 &&
            mppi_set_tuning(ctrl.handle(), MPPI_TUNE_ARMED_US, atoi(argv[6] + 6)) != MPPI_OK) {
            fprintf(stderr, "host_loop: %s\n", mppi_last_error(ctrl.handle()));
            return 4;
        }
        vector<float> state(sDim, 0.f), action(aDim, 0.f);
        float d2_init = 0;
        for (int i = 0; i < sDim; i++) d2_init += (state[i] - goal[i]) * (state[i] - goal[i]);
        bool done(false);
        int it = 0;
        while (!done) {
            action = ctrl.next(state);
            for (int j = 0; j < aDim; j++) { // env.simulate(action); env.get_x(state)
                state[2 * j] = state[2 * j] + dt * state[2 * j + 1] + (dt * dt / 2.f) * action[j];
                state[2 * j + 1] = state[2 * j + 1] + dt * action[j];
            }
            ctrl.saveNext(state);
            done = ++it >= steps;
        }
        if (argc > 5 && std::string(argv[5]) != "-") ctrl.toCSV(argv[5]);
        float d2 = 0;
        for (int i = 0; i < sDim; i++) d2 += (state[i] - goal[i]) * (state[i] - goal[i]);
        printf("after %d closed-loop steps: |x-goal|^2 = %g\n", steps, d2);
        // Record start time (main.cpp:55-64, un-commented)
        auto start = chrono::high_resolution_clock::now();
        for (int i = 0; i < 100; i++) ctrl.next(state);
        auto finish = chrono::high_resolution_clock::now();
        chrono::duration<double> elapsed = finish - start;
        printf("Execution time: %g ms / control step (K=%d tau=%d s=%d a=%d) = %g rollouts/s\n",
               elapsed.count() / 100. * 1e3, k, tau, sDim, aDim, k / (elapsed.count() / 100.));
        return d2 < 0.5f * d2_init ? 0 : 1; // the loop must at least halve the squared distance to the goal
    } catch (const std::exception &e) {
        fprintf(stderr, "host_loop: %s\n", e.what());
        return 3;
    }
}
