#!/usr/bin/env python3
"""The reference's Python entry point (scripts/main.py:13-121) against the MI355X control step:
   python examples/main.py --new --config examples/config/point_mass3d.yaml --task examples/config/static_task3d.yaml -s 200 -l --log_dir runs/a
   python examples/main.py --replay --log_dir runs/a/controller
   python examples/main.py --new --config examples/config/uuv_sim.yaml --model examples/config/rexrov2.yaml --task examples/config/static_quat_task.yaml -s 50
parses the same YAML keys, builds Simulation / PointMassModel / StaticCost / ControllerBase with the
reference's constructor arguments, runs the closed loop, and writes the transition log as CSV.
--new / --replay are the reference's mutually exclusive modes (main.py:17-24): with -l the run's config and task are
dumped as <log_dir>/controller/{config,task}.yaml (what ObserverBase does, observer_base.py:39-54), and --replay reads
exactly those two files back (utile.py:53-59 parse_dir) and repeats the experiment — same seed, same Philox stream,
so the replayed transitions are bit-identical."""
import argparse
import glob
import os
import sys
import time

import numpy as np
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mppi_tf_amd as m  # noqa: E402
from mppi_tf_amd.simulation import Simulation  # noqa: E402


def parse_config(path):  # scripts/src/misc/utile.py:41-44
    with open(path) as fh:
        return yaml.safe_load(fh)


def parse_dir(log_dir):  # scripts/src/misc/utile.py:47-59: the config.yaml / task.yaml a logged run left behind
    config_file = task_file = None
    for name in os.listdir(log_dir):
        if not os.path.isfile(os.path.join(log_dir, name)):
            continue
        if name == "config.yaml":
            config_file = os.path.join(log_dir, name)
        elif name == "task.yaml":
            task_file = os.path.join(log_dir, name)
    if config_file is None or task_file is None:
        raise FileNotFoundError("%s holds no config.yaml / task.yaml (a run logged with -l writes them)" % log_dir)
    return parse_config(config_file), task_file


def write_log_dir(log_path, conf, task):  # observer_base.py:39-54
    logdir = os.path.join(log_path, "controller")
    os.makedirs(logdir, exist_ok=True)
    for name, d in (("config.yaml", conf), ("task.yaml", task)):
        with open(os.path.join(logdir, name), "w") as fh:
            yaml.dump(d, fh)
    return logdir


def get_cost(task_file, lam, gamma, upsilon, sigma):  # scripts/src/cost.py:51-64
    task = parse_config(task_file)
    sigma = np.asarray(sigma, np.float32)
    if task["type"] == "static":  # cost.py:7-11
        goal = np.asarray(task["goal"], np.float32).reshape(-1, 1)
        return m.StaticCost(lam, gamma, upsilon, sigma, goal, np.asarray(task["Q"], np.float32), diag=bool(task.get("diag", False)))
    if task["type"] == "static_quat":  # cost.py:13-17: 13-state goal, Q over (position, attitude angle, velocities) = 10
        goal = np.asarray(task["goal"], np.float32).reshape(-1, 1)
        return m.StaticQuatCost(lam, gamma, upsilon, sigma, goal, np.asarray(task["Q"], np.float32), diag=bool(task.get("diag", False)))
    if task["type"] == "elipse":  # cost.py:21-30 — which hands task['center_y'] to center_x and vice versa; reproduced
        center_y, center_x = task["center_x"], task["center_y"]
        return m.ElipseCost(lam, gamma, upsilon, sigma, task["a"], task["b"], center_x, center_y, task["speed"],
                            task["m_state"], task["m_vel"])
    if task["type"] == "elipse3d":
        # cost.py:33-42 still calls ElipseCost3D with the 2-D class's argument list (a, b, centre, depth -10, speed, ...), which the class
        # (elipse_cost.py:101-122: normal, aVec, axis, center, speed, v_speed, mState, mVel) no longer takes. Read as what it means:
        # an ellipse with half-axes a, b in the horizontal plane at depth -10, or the class's own keys when the task file has them.
        center_y, center_x = task["center_x"], task["center_y"]
        return m.ElipseCost3D(lam, gamma, upsilon, sigma, np.asarray(task.get("normal", [0.0, 0.0, 1.0]), np.float32).reshape(3, 1),
                              np.asarray(task.get("aVec", [1.0, 0.0, 0.0]), np.float32).reshape(3, 1),
                              np.asarray(task.get("axis", [task["a"], task["b"]]), np.float32).reshape(2, 1),
                              np.asarray(task.get("center", [center_x, center_y, task.get("depth", -10.0)]), np.float32).reshape(3, 1),
                              task["speed"], task.get("v_speed", 0.0), task["m_state"], task["m_vel"])
    raise NotImplementedError("cost types on the accelerated path: static, static_quat, elipse, elipse3d (waypoints: broken in the reference itself)")


def get_model(model_dict, samples, dt, state_dim, action_dim, name="model"):  # scripts/src/model.py:52-66
    kind = model_dict["type"]
    if kind == "point_mass":
        return m.PointMassModel(model_dict["mass"], dt, state_dim, action_dim)
    if kind == "auv":  # model.py:41-49
        return m.AUVModel(modelDict=model_dict, inertialFrameId=model_dict.get("frame_id", "world"), actionDim=action_dim, name=name, k=samples,
                          dt=dt, parameters=model_dict)
    if kind in ("auv_nn", "auv_nn_speed"):  # model.py:16-32: random initial weights (no Keras weight file here): --train makes them useful
        cls = m.NNAUVModel if kind == "auv_nn" else m.NNAUVModelSpeed
        return cls(modelDict=model_dict, k=samples, stateDim=state_dim, actionDim=action_dim, mask=np.array(model_dict["mask"]) if "mask" in model_dict else None,
                   dt=dt)
    raise NotImplementedError("model types of this entry point: point_mass, auv, auv_nn, auv_nn_speed (neural_net: NNModel.build_step_graph "
                              "is NotImplemented in the reference, nn_model.py:101-117)")


class AUVSimulation:
    """Host-loop plant for the 13-state family: the Fossen model itself, one device step per control step (the reference drives
    uuv_simulator over ROS, which is not part of this build). Same getState / step / getTime surface as Simulation."""

    def __init__(self, model, x0=None, dt=0.1):
        self.model, self.dt, self.time = model, float(dt), 0.0
        self.x = np.asarray(x0 if x0 is not None else [0, 0, 0, 0, 0, 0, 1] + [0] * 6, np.float32).reshape(13, 1)

    def getTime(self):
        return self.time

    def getState(self):
        return self.x.copy()

    def step(self, u, goal=None):
        self.x = np.asarray(self.model.predict(self.x[None], np.asarray(u, np.float32).reshape(1, -1, 1)), np.float32).reshape(13, 1)
        self.time += self.dt
        return self.getState()


def main():
    ap = argparse.ArgumentParser(prog="mppi", description="mppi on MI355X")
    group = ap.add_mutually_exclusive_group(required=True)  # main.py:17-24
    group.add_argument("--replay", action="store_true", help="replay the experiment a log dir describes")
    group.add_argument("--new", action="store_true", help="expects a config and a task file")
    ap.add_argument("--log_dir", type=str, help="--replay: the logged run's directory (holding config.yaml and task.yaml); "
                                                "--new -l: where <log_dir>/controller/ is created")
    ap.add_argument("--config", type=str)
    ap.add_argument("--task", type=str)
    ap.add_argument("--model", type=str, default=None, help="model description file (config/models/*.yaml: type point_mass | auv); "
                                                              "default: the point mass of --config")
    ap.add_argument("-l", "--log", action="store_true", help="leave config.yaml / task.yaml (and the CSV) in the log dir")
    ap.add_argument("-s", "--steps", type=int, default=200)
    ap.add_argument("--plant", type=str, default=None, help="model description of the PLANT when --model is a learned one (type auv); "
                                                              "the controller then learns it online")
    ap.add_argument("-t", "--train", type=int, default=0, help="train the learned model on the replay buffer every this many steps "
                                                               "(main.py:51-52, 104-105) and push the weights into the controller; 0: never")
    ap.add_argument("--csv", type=str, default=None, help="write the (x, u, x_next) log here (DataBase::toCSV's bytes)")
    args = ap.parse_args()
    if args.new:
        if not args.config or not args.task:
            ap.error("--new expects --config and --task")
        conf = parse_config(args.config)
    else:
        if not args.log_dir:
            ap.error("--replay expects --log_dir")
        conf, args.task = parse_dir(args.log_dir)
    if args.log:
        if not args.log_dir:
            ap.error("-l expects --log_dir")
        logdir = write_log_dir(args.log_dir, conf, parse_config(args.task))
        if args.csv is None:
            args.csv = os.path.join(logdir, "transitions.csv")
    model_dict = parse_config(args.model) if args.model else conf.get("model") or {"type": "point_mass", "mass": conf.get("mass", 1.0)}
    if args.log and args.model:
        conf = dict(conf, model=model_dict)  # a replay finds the model in the dumped config
        write_log_dir(args.log_dir, conf, parse_config(args.task))
    learned = model_dict["type"] in ("auv_nn", "auv_nn_speed")
    auv = model_dict["type"] == "auv" or learned
    s_dim, a_dim = conf.get("state-dim", 13 if auv else None), conf.get("action-dim", 6 if auv else None)
    model = get_model(model_dict, conf["samples"], conf["dt"], s_dim, a_dim)
    if learned:
        if not args.plant:
            ap.error("a learned model needs --plant (the system it controls and learns)")
        plant = get_model(parse_config(args.plant), 1, conf["dt"], s_dim, a_dim, name="plant")
        learner = m.LearnerBase(model, bufferSize=max(args.steps, 1), logPath=os.path.join(args.log_dir, "learner") if args.log_dir else None)
        trained_steps = 0
        if args.train and args.log_dir:  # resume: the newest weights_step<N> a previous --train run left (weights, normalisation, Adam state)
            saved = sorted((f for f in glob.glob(os.path.join(learner.logdir, "weights_step*")) if f.rsplit("step", 1)[1].isdigit()),
                           key=lambda f: int(f.rsplit("step", 1)[1]))
            if saved:
                learner.load_params(saved[-1])
                learner.step = trained_steps = int(saved[-1].rsplit("step", 1)[1])  # the run's count of training epochs (train_all starts a new Adam every round)
                print("resumed the learned model from %s (%d training epochs so far)" % (saved[-1], trained_steps))
    sim = AUVSimulation(plant if learned else model, conf.get("x0"), conf["dt"]) if auv else Simulation(conf.get("env"), s_dim, a_dim, None, False,
                                                                                                         dt=conf["dt"], mass=model_dict.get("mass", 1.0))
    cost = get_cost(args.task, conf["lambda"], conf.get("gamma", 1.0), conf.get("upsilon", 1.0), conf["noise"])
    cont = m.ControllerBase(model, cost, k=conf["samples"], tau=conf["horizon"], sDim=s_dim, aDim=a_dim,
                            lam=conf["lambda"], upsilon=conf.get("upsilon", 1.0), sigma=np.asarray(conf["noise"], np.float32))
    ts = []
    for step in range(args.steps):
        x = sim.getState()
        t0 = time.perf_counter()
        u = cont.next(x)
        ts.append(time.perf_counter() - t0)
        x_next = sim.step(u)
        cont.save(x, u, x_next)
        if learned:
            learner.add_rb(x[None], np.asarray(u, np.float32).reshape(1, -1, 1), x_next[None])
            if args.train and (step + 1) % args.train == 0:  # main.py:104-105; the weights then travel into the live controller
                learner.stats()
                first, last = learner.train_all(learningRate=3e-3, epoch=200)
                cont.update_model()
                print("step %d: trained on %d transitions, normalised loss %.4f -> %.4f" % (step + 1, step + 1, first, last))
                if args.log_dir:  # learner_base.py:66-68 save_params
                    print("saved", learner.save_params(learner.step))
    steady = np.sort(ts[min(5, len(ts) - 1):])  # the first calls load the code objects
    goal_of = getattr(cost, "getGoal", None) or getattr(cost, "get_goal", None)
    if goal_of is not None and auv:  # StaticCost on 13 states / StaticQuatCost: distance in position
        tail = "|p - goal_p| = %.4f m" % float(np.linalg.norm(sim.getState().ravel()[:3] - np.asarray(goal_of()).ravel()[:3]))
    elif goal_of is not None:
        tail = "|x - goal| = %.4f" % float(np.linalg.norm(sim.getState().ravel() - np.asarray(goal_of()).ravel()))
    elif hasattr(cost, "position_error"):  # ElipseCost3D
        st = sim.getState()[None]
        tail = "elipse3d state cost %.4f, speed %.3f m/s" % (float(np.ravel(cost.state_cost("s", st))[0]), float(np.linalg.norm(st[0, 7:10, 0])))
    else:
        d = cost.dist(sim.getState())
        tail = "elipse x_dist = %.4f v_dist = %.4f" % (float(d["x_dist"]), float(d["v_dist"]))
    print("%d control steps, controller median %.3f ms/step (first call %.1f ms), %s" % (
        args.steps, 1e3 * float(np.median(steady)), 1e3 * ts[0], tail))
    if args.csv:
        cont._h.to_csv(args.csv)
        print("wrote", args.csv)


if __name__ == "__main__":
    main()
