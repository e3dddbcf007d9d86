"""MuJoCo-free stand-in for the reference's plant wrappers (the CALLER of the control step):
   scripts/src/mujoco/simulation.py  Simulation.getTime / getGoal / getState / step(u)
   src/mj_pm_env.cpp:118-201         PointMassEnv::simulate / step / get_x
The reference steps envs/point_mass{1,2,3}d.xml (slide joints + motors) in MuJoCo, which is not
available here; this plant integrates the same point mass the controller's ModelBase assumes
(x' = A x + (B/m) u per axis), optionally with the 3d env's joint damping. It is environment
code on the host, not part of the accelerated path."""
import numpy as np


class Simulation:
    def __init__(self, xml_file=None, s_dim=2, a_dim=1, goal=None, render=False, dt=0.1, mass=1.0, damping=0.0):
        if render:
            raise NotImplementedError("no viewer: MuJoCo/GLFW are not part of this build")
        self.s_dim, self.a_dim, self.dt, self.mass, self.damping = int(s_dim), int(a_dim), float(dt), float(mass), float(damping)
        self.goal = goal
        self.x = np.zeros((self.s_dim, 1), np.float32)
        self.time = 0.0

    def getTime(self):
        return self.time

    def getGoal(self):
        return self.goal

    def getState(self):
        return self.x.copy()

    def step(self, u, goal=None):
        u = np.asarray(u, np.float32).reshape(-1)
        dt, m = np.float32(self.dt), np.float32(self.mass)
        for j in range(self.a_dim):
            p, q = self.x[2 * j, 0], self.x[2 * j + 1, 0]
            f = u[j] - np.float32(self.damping) * q
            self.x[2 * j, 0] = p + dt * q + (dt * dt / np.float32(2)) / m * f
            self.x[2 * j + 1, 0] = q + dt / m * f
        self.time += self.dt
        return self.getState()


class PointMassEnv:
    """C++ spelling (include/mj_env.hpp:20-33): simulate(u) -> done, step(x, u), get_x(x)."""

    def __init__(self, s_dim, a_dim, dt=0.1, mass=1.0, simend=20.0):
        self._sim = Simulation(None, s_dim, a_dim, dt=dt, mass=mass)
        self._simend = simend

    def simulate(self, u):
        if self._sim.getTime() < self._simend:
            self._sim.step(u)
            return False
        return True

    def step(self, x, u):
        x[:] = self._sim.step(u).ravel().tolist()

    def get_x(self, x):
        x[:] = self._sim.getState().ravel().tolist()
