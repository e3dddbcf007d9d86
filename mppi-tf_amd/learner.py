"""Host-side mirror of the reference's learner for the learned model_base:
scripts/src/learners/learner_base.py LearnerBase — replay buffer, normalisation statistics, train_all / train / _train_step.

Same method names and argument meaning. What is numeric — the network's forward pass, the loss, the gradients and the Adam
update — runs on the GPU (mppi_learner.hip through the C-ABI's mppi_learner_*); the bookkeeping around it (replay buffer,
mean / std of the data, noise augmentation of the inputs) is numpy on the host, as it is host-side Python in the reference.
The reference's replay buffer is cpprb's ReplayBuffer (a ring of transitions): restated here as three numpy arrays.
TensorBoard logging, plotting, k-fold grid search are outside the path.
"""
import os

import numpy as np

from ._lib import Learner


class ReplayBuffer:
    """a ring of (obs, act, next_obs) transitions (cpprb.ReplayBuffer's surface as learner_base.py:32-63 uses it)"""

    def __init__(self, size, sDim, aDim):
        self.size, self.n, self.head = int(size), 0, 0
        self.obs, self.act, self.next_obs = (np.zeros((self.size, d, 1)) for d in (sDim, aDim, sDim))

    def add(self, obs, act, next_obs):
        obs, act, next_obs = (np.asarray(v, np.float64) for v in (obs, act, next_obs))
        if obs.ndim == 2:
            obs, act, next_obs = obs[None], act[None], next_obs[None]
        for o, a, n in zip(obs, act, next_obs):
            self.obs[self.head], self.act[self.head], self.next_obs[self.head] = o, a, n
            self.head = (self.head + 1) % self.size
            self.n = min(self.n + 1, self.size)

    def get_all_transitions(self):
        return dict(obs=self.obs[:self.n], act=self.act[:self.n], next_obs=self.next_obs[:self.n])

    def save_transitions(self, filename):
        np.savez(filename, **self.get_all_transitions())

    def load_transitions(self, filename):
        d = np.load(filename)
        self.add(d["obs"], d["act"], d["next_obs"])


class Adam:
    """the hyper-parameters of tf.optimizers.Adam (learner_base.py:31, :149); the state lives on the device"""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon


class LearnerBase:
    def __init__(self, model, filename=None, bufferSize=264, numEpochs=100, batchSize=30, log=False, logPath=None, device=0):
        self.model = model
        self.sDim, self.aDim = model.get_state_dim(), model.get_action_dim()
        self.optimizer = Adam(learning_rate=0.5)  # learner_base.py:31
        self.rb = ReplayBuffer(bufferSize, self.sDim, self.aDim)
        self.numEpochs, self.batchSize = numEpochs, batchSize
        if filename is not None:
            self.load_rb(filename)
        self.log, self.step, self.sigma = log, 0, 0.001
        self._device, self._dev, self._dev_model = device, None, None
        self.last_losses = []
        self.logdir = logPath  # where save_params puts weights_step<N> (learner_base.py:40-52 stamps a sub-directory; this keeps the path given)
        self.val_log = []      # (epoch, validate(...)) of the last train(..., val=...)

    # ---- replay buffer (learner_base.py:54-66)
    def load_rb(self, filename):
        self.rb.load_transitions(filename)

    def add_rb(self, x, u, xNext):
        self.rb.add(obs=x, act=u, next_obs=xNext)

    def rb_trans(self):
        return {k: v.copy() for k, v in self.rb.get_all_transitions().items()}

    def save_rb(self, filename):
        self.rb.save_transitions(filename)

    # ---- persistence (learner_base.py:66-68 save_params -> NNModel.save_params, nn_model.py:137-142). The reference saves the
    # Keras model only; this file also carries both Adam moments and the step count (mppi_learner_save's flat format), so that a
    # run resumed in a new process continues bit for bit.
    def params_path(self, step, logdir=None):
        return os.path.join(logdir or self.logdir or ".", "weights_step{}".format(step))

    def save_params(self, step, logdir=None):
        path = self.params_path(step, logdir)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        self._device_learner(self.model).save(path, self.model.normalisation())
        return path

    def load_params(self, path):
        """weights + normalisation into the model, weights + Adam moments + step count into the device learner"""
        self._dev, norm = Learner.from_file(path, device=self._device)
        self._dev_model = self.model
        self._pull(self.model)
        if norm is not None:
            self.model.set_Xmean_Xstd(norm["xmean"], norm["xstd"])
            self.model.set_Ymean_Ystd(norm["ymean"], norm["ystd"])
        self.step = self._dev.step_count()

    def stats(self):
        """mean / std of the un-normalised training pairs become the model's normalisation (learner_base.py:71-83)"""
        data = self.rb_trans()
        X, y = self.model.prepare_training_data(data["obs"], data["next_obs"], data["act"], norm=False)
        std = lambda a: np.where(np.std(a, axis=0) > 0, np.std(a, axis=0), 1.0)  # a constant feature keeps its scale (the reference divides by 0)
        self.model.set_Xmean_Xstd(np.mean(X, axis=0), std(X))
        self.model.set_Ymean_Ystd(np.mean(y, axis=0), std(y))

    # ---- training (learner_base.py:146-153, 324-358, 469-496)
    def _device_learner(self, model):
        """The device learner holding `model`'s CURRENT weights. In the reference learner and model share tf.Variables, so
        model.update_weights(...) (the grid search's reset, learner_base.py:143; a loaded weight file) takes effect on the next
        step; here the weights are pushed before every train / evaluate — a few KB (they used to be uploaded once,
        at first use, and a later update_weights was silently ignored and then overwritten by _pull). The Adam moments stay, as
        an optimizer's slots do when its variables are assigned. One device learner per model object."""
        w = model.get_weights()
        weights = dict(W=w[0::2], b=w[1::2])
        if self._dev is None or self._dev_model is not model:
            self._dev, self._dev_model = Learner(weights, device=self._device), model
        else:
            self._dev.set_weights(weights)
        return self._dev

    def train_all(self, learningRate=0.1, batchSize=32, epoch=100, val=None, writer=None, augment=True):
        self.optimizer = Adam(learning_rate=learningRate)
        if self._dev is not None:
            self._dev.reset_optimizer()  # a new tf.optimizers.Adam (learner_base.py:149)
        data = self.rb_trans()
        X, y = self.model.prepare_training_data(data["obs"], data["next_obs"], data["act"])
        return self.train(X, y, epoch=epoch, learningRate=learningRate, val=val, writer=writer, augment=augment)

    def augment_data(self, X, y, samples=5, sigma=0.001, seed=1):
        """every pair `samples` times with N(0, sigma) noise on the inputs (learner_base.py:446-466). The reference then passes the
        DE-normalised arrays to a step that expects normalised ones (:464-465, a no-op at identity statistics): not reproduced —
        the network is trained on the normalised pairs it will be evaluated on."""
        XAug, yAug = np.repeat(X, samples, axis=0), np.repeat(y, samples, axis=0)
        return XAug + np.random.default_rng(seed).normal(0.0, sigma, XAug.shape), yAug

    def train(self, X, y, epoch=1, learningRate=0.1, augment=False, val=None, writer=None):
        """`epoch` full-batch Adam steps. With augmentation the reference draws fresh input noise every epoch (learner_base.py:326-329,
        with the SAME seed every time: the same noise); here the augmented set is drawn once and stays on the device."""
        XEp, yEp = self.augment_data(X, y, sigma=self.sigma) if augment else (X, y)
        dev = self._device_learner(self.model)
        dev.set_data(XEp, yEp)
        o = self.optimizer
        self.val_log = []
        if val is None:
            first, last = dev.train(epoch, learningRate, o.beta_1, o.beta_2, o.epsilon)
        else:
            # learner_base.py:336-358: every 10th epoch the model is validated on held-out trajectories val = (gtTrajs, actionSeqs):
            # the epochs run on the device in blocks of 10 with the validation between them (the reference logs the numbers to
            # TensorBoard; they are kept in self.val_log and handed to `writer(epoch, dict)` if one is given)
            first = last = None
            for e0 in range(0, epoch, 10):
                self._pull(self.model)
                errs = self.validate(self.model, val[1], val[0], transition=True, split=True, norm=True)
                names = ("Val-Total", "Val-Total-Norm", "Val-Split", "Val-Split-Norm", "Val-Transition", "Val-Transition-Norm",
                         "Val-Transition-Split", "Val-Transition-Split-Norm")
                entry = dict(zip(names, errs))
                self.val_log.append((e0, entry))
                if callable(writer):
                    writer(e0, entry)
                f, last = dev.train(min(10, epoch - e0), learningRate, o.beta_1, o.beta_2, o.epsilon)
                first = f if first is None else first
        self.last_losses = [first, last]
        self._pull(self.model)
        self.step += epoch
        return first, last

    # ---- validation (learner_base.py:218-320)
    def norm_trajs(self, model, trajs):
        """trajectories [k, tau, s] in the network's output units (learner_base.py:360-367: (traj - Ymean) / Ystd)"""
        return (np.asarray(trajs, np.float64) - np.asarray(model.Ymean)) / np.asarray(model.Ystd)

    def validate(self, model, actionSeqs, gtTrajs, plot=False, transition=False, split=False, norm=False):
        """Error of the model on k trajectories gtTrajs [k, tau, s] with their action sequences actionSeqs [k, tau, a]
        (learner_base.py:218-320): the model is rolled tau-1 steps from each trajectory's first state (on the device, the model's own
        step) and compared with the ground truth — err [, errNorm] [, errSplit, errSplitNorm] — and with transition=True every
        (x_t, u_t) -> x_t+1 pair is predicted on its own (transErr...). Same return tuples as the reference for every flag combination."""
        actionSeqs, gtTrajs = np.asarray(actionSeqs, np.float64), np.asarray(gtTrajs, np.float64)
        tau = actionSeqs.shape[1]
        state = gtTrajs[:, 0][..., None]
        trajs = [state[:, None]]
        for i in range(tau - 1):
            state = np.asarray(model.build_step_graph("val", state, actionSeqs[:, i][..., None]), np.float64)
            trajs.append(state[:, None])
        trajs = np.concatenate(trajs, axis=1)[..., 0]
        errSplit = np.mean((trajs - gtTrajs) ** 2, axis=(0, 1))
        err = float(np.mean(errSplit))
        out_n = len(np.asarray(model.Ymean).ravel())
        if out_n == trajs.shape[-1]:
            errSplitNorm = np.mean((self.norm_trajs(model, trajs) - self.norm_trajs(model, gtTrajs)) ** 2, axis=(0, 1))
        else:  # NNAUVModelSpeed normalises the 6 velocity outputs only: its trajectory error in those units (state[7:13])
            errSplitNorm = np.mean(((trajs[..., -out_n:] - gtTrajs[..., -out_n:]) / np.asarray(model.Ystd)) ** 2, axis=(0, 1))
        errNorm = float(np.mean(errSplitNorm))
        if transition:
            k, s_dim, a_dim = gtTrajs.shape[0], gtTrajs.shape[2], actionSeqs.shape[2]
            actions = actionSeqs[:, :-1].reshape((tau - 1) * k, a_dim, 1)
            states = gtTrajs[:, :-1].reshape((tau - 1) * k, s_dim, 1)
            nextStates = gtTrajs[:, 1:].reshape((tau - 1) * k, s_dim, 1)
            XNorm, yNorm = model.prepare_training_data(states, nextStates, actions)
            dev = Learner(dict(W=model.get_weights()[0::2], b=model.get_weights()[1::2]), device=self._device)  # the network alone (model._predict_nn)
            dev.set_data(XNorm, yNorm)
            _, predNorm = dev.evaluate(pred=True)
            dev.close()
            yN = np.asarray(yNorm, np.float32)
            transErrSplitNorm = np.mean((predNorm - yN) ** 2, axis=0)
            transErrNorm = float(np.mean(transErrSplitNorm))
            transErrSplit = np.mean(((predNorm - yN) * np.asarray(model.Ystd)) ** 2, axis=0)  # denormalizeY: Ymean cancels in the difference
            transErr = float(np.mean(transErrSplit))
        # What the caller unpacks (learner_base.py:301-322), built instead of enumerated: the trajectory error, then with `norm` its normalised
        # twin, then with `split` the per-dimension pair; with `transition` the same pattern again for the one-step error.
        def pattern(e, eNorm, eSplit, eSplitNorm):
            out = [e] + ([eNorm] if norm else [])
            if split:
                out += [eSplit] + ([eSplitNorm] if norm else [])
            return out

        out = pattern(err, errNorm, errSplit, errSplitNorm)
        if transition:
            out += pattern(transErr, transErrNorm, transErrSplit, transErrSplitNorm)
        return out[0] if len(out) == 1 else tuple(out)

    def _train_step(self, model, optimizer, Xnorm, Ynorm, split=False, norm=False):
        """ONE Adam step on (Xnorm, Ynorm) (learner_base.py:469-496) -> (loss, grads): the loss of the forward pass before the update in
        normalised AND de-normalised units is the same number at identity statistics; returned as (loss, lossNorm, grads) with norm=True."""
        dev = self._device_learner(model)
        dev.set_data(Xnorm, Ynorm)
        lossNorm, grads, pred = dev.evaluate(grads=True, pred=True)
        dev.train(1, optimizer.learning_rate, optimizer.beta_1, optimizer.beta_2, optimizer.epsilon)
        self._pull(model)
        # the de-normalised loss (learner_base.py:482-486): Ymean cancels in the difference, Ystd scales it
        loss = float(np.mean(((pred - np.asarray(Ynorm, np.float32)) * np.asarray(model.Ystd, np.float64)) ** 2))
        glist = []
        for w, b in zip(grads["W"], grads["b"]):
            glist += [w, b]
        if norm:
            return loss, lossNorm, glist
        return loss, glist

    def evaluate(self, model, X, y):
        dev = self._device_learner(model)
        dev.set_data(X, y)
        return dev.evaluate()

    def _pull(self, model):
        w = self._dev.get_weights()
        flat = []
        for W, b in zip(w["W"], w["b"]):
            flat += [W, b]
        model.update_weights(flat, msg=False)
