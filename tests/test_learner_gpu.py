"""The learner of the learned model_base on the GPU (csrc/mppi_learner.hip; LearnerBase.train / _train_step,
scripts/src/learners/learner_base.py:324-358, 469-496) against a torch-CPU fp32 reference of the same update: autograd for the
gradients of mean((nn(X) - Y)^2), tf.keras.optimizers.Adam's update written out in numpy for the step. Then the loop the
reference closes: transitions of the (device) Fossen AUVModel -> LearnerBase(NNAUVModel) -> trained weights -> the NNAUVModel
controller rolls them out."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as g
    g.build()
    import mppi_tf_amd as mod
    return mod


def make_net(dims, seed=0):
    rng = np.random.default_rng(seed)
    return dict(W=[(rng.uniform(-1, 1, (dims[i], dims[i + 1])) * np.sqrt(6.0 / (dims[i] + dims[i + 1]))).astype(F32) for i in range(len(dims) - 1)],
                b=[(0.1 * rng.standard_normal(dims[i + 1])).astype(F32) for i in range(len(dims) - 1)])


def torch_loss_and_grads(net, X, Y):
    import torch
    Ws = [torch.tensor(w, requires_grad=True) for w in net["W"]]
    bs = [torch.tensor(b, requires_grad=True) for b in net["b"]]
    h = torch.tensor(X)
    for l, (W, b) in enumerate(zip(Ws, bs)):
        h = h @ W + b
        if l + 1 < len(Ws):
            h = torch.relu(h)
    loss = torch.mean((h - torch.tensor(Y)) ** 2)
    loss.backward()
    return float(loss.detach()), [w.grad.numpy() for w in Ws], [b.grad.numpy() for b in bs]


def keras_adam(net, state, gW, gb, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam: m, v, lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), w -= lr_t m / (sqrt(v) + eps); fp32"""
    lr_t = F32(lr) * np.sqrt(F32(1) - F32(b2) ** F32(t)) / (F32(1) - F32(b1) ** F32(t))
    for key, grads in (("W", gW), ("b", gb)):
        for l, g in enumerate(grads):
            mm, vv = state[key][l]
            mm[:] = F32(b1) * mm + F32(1 - b1) * g
            vv[:] = F32(b2) * vv + F32(1 - b2) * g * g
            net[key][l] = (net[key][l] - lr_t * mm / (np.sqrt(vv) + F32(eps))).astype(F32)


@pytest.mark.parametrize("dims,n", [([16, 32, 32, 32, 13], 1000), ([9, 16, 6], 333), ([16, 32, 13], 9001), ([4, 8, 8, 8, 2], 64)])
def test_gradients_match_torch_autograd(m, dims, n):
    """loss and dLoss/dW, dLoss/db of the reference's network shapes (ragged n: partial last block, odd sample count, more than one
    4096-sample chunk of the batch GEMM) against torch CPU fp32 autograd."""
    net = make_net(dims, 1)
    rng = np.random.default_rng(2)
    X, Y = rng.standard_normal((n, dims[0])).astype(F32), rng.standard_normal((n, dims[-1])).astype(F32)
    lr = m.Learner(net)
    lr.set_data(X, Y)
    loss, g, pred = lr.evaluate(grads=True, pred=True)
    tl, tW, tb = torch_loss_and_grads(net, X, Y)
    assert abs(loss - tl) <= 2e-6 * abs(tl)
    for l in range(len(dims) - 1):
        scale = max(np.abs(tW[l]).max(), 1e-12)
        np.testing.assert_allclose(g["W"][l], tW[l], rtol=2e-5, atol=2e-6 * scale)
        np.testing.assert_allclose(g["b"][l], tb[l], rtol=2e-5, atol=2e-6 * max(np.abs(tb[l]).max(), 1e-12))
    assert pred.shape == (n, dims[-1]) and abs(np.mean((pred - Y) ** 2) - tl) <= 1e-5 * tl


def test_adam_steps_match_the_keras_update(m):
    """40 full-batch Adam steps on the device against the same 40 steps done with torch autograd + the Keras update in numpy:
    weights agree to 2e-5 absolute (they move by ~0.04 in total), the loss falls, two learners give identical bits."""
    dims, n, steps, lr_rate = [16, 32, 32, 32, 13], 2000, 40, 1e-3
    net = make_net(dims, 3)
    rng = np.random.default_rng(4)
    X = rng.standard_normal((n, 16)).astype(F32)
    Y = (np.tanh(X @ rng.standard_normal((16, 13)) * 0.3)).astype(F32)
    a, b = m.Learner(net), m.Learner(net)
    a.set_data(X, Y)
    b.set_data(X, Y)
    first, last = a.train(steps, lr_rate)
    b.train(steps // 2, lr_rate)
    b.train(steps - steps // 2, lr_rate)  # the optimiser state persists across calls
    wa, wb = a.get_weights(), b.get_weights()
    for l in range(4):
        np.testing.assert_array_equal(wa["W"][l], wb["W"][l])
        np.testing.assert_array_equal(wa["b"][l], wb["b"][l])
    assert a.step_count() == steps and last < first
    ref = dict(W=[w.copy() for w in net["W"]], b=[v.copy() for v in net["b"]])
    state = dict(W=[(np.zeros_like(w), np.zeros_like(w)) for w in ref["W"]], b=[(np.zeros_like(v), np.zeros_like(v)) for v in ref["b"]])
    losses = []
    for t in range(1, steps + 1):
        tl, gW, gb = torch_loss_and_grads(ref, X, Y)
        losses.append(tl)
        keras_adam(ref, state, gW, gb, t, lr_rate)
    assert abs(first - losses[0]) <= 2e-6 * losses[0] and abs(last - losses[-1]) <= 2e-4 * losses[-1]
    moved = max(np.abs(wa["W"][l] - net["W"][l]).max() for l in range(4))
    # Adam normalises by sqrt(v): where a gradient is of the order of eps (dead relu units) its relative rounding error IS the update's,
    # so 40 steps of two fp32 evaluations drift apart by up to ~1 % of the distance moved; the update RULE itself is checked exactly below
    for l in range(4):
        np.testing.assert_allclose(wa["W"][l], ref["W"][l], rtol=0, atol=2e-2 * moved)
        np.testing.assert_allclose(wa["b"][l], ref["b"][l], rtol=0, atol=2e-2 * moved)
    # two consecutive steps from the device's OWN gradients: the Keras update in numpy reproduces the device's weights to fp32 rounding
    c = m.Learner(net)
    c.set_data(X, Y)
    chk = dict(W=[w.copy() for w in net["W"]], b=[v.copy() for v in net["b"]])
    st2 = dict(W=[(np.zeros_like(w), np.zeros_like(w)) for w in chk["W"]], b=[(np.zeros_like(v), np.zeros_like(v)) for v in chk["b"]])
    for t in (1, 2):
        _, g = c.evaluate(grads=True)
        c.train(1, lr_rate)
        keras_adam(chk, st2, g["W"], g["b"], t, lr_rate)
        wc = c.get_weights()
        for l in range(4):
            np.testing.assert_allclose(wc["W"][l], chk["W"][l], rtol=0, atol=3e-7)
            np.testing.assert_allclose(wc["b"][l], chk["b"][l], rtol=0, atol=3e-7)
    print("Adam: loss %.5f -> %.5f; weights moved by up to %.3g, device vs reference differ by up to %.3g"
          % (first, last, moved, max(np.abs(wa["W"][l] - ref["W"][l]).max() for l in range(4))))
    assert moved > 1e-2
    a.reset_optimizer()
    assert a.step_count() == 0


def test_learner_argument_errors(m):
    with pytest.raises(m.MppiError) as e:
        m.Learner(make_net([16, 64, 13]))
    assert e.value.status == 4
    lr = m.Learner(make_net([4, 8, 2]))
    with pytest.raises(m.MppiError):
        lr.train(5, 1e-3)  # no data
    lr.set_data(np.zeros((10, 4), F32), np.zeros((10, 2), F32))
    with pytest.raises(m.MppiError):
        lr.train(0, 1e-3)


def test_train_the_auv_network_on_the_fossen_model_and_control_with_it(m):
    """The loop the reference closes (main.py with a learnable model): transitions of the Fossen AUVModel (rolled on the device) fill
    the replay buffer, LearnerBase computes the normalisation statistics and trains NNAUVModel's Dense(32)x3 network with Adam on
    the device; the trained model predicts the plant's next state far better than the untrained one, and the NNAUVModel controller
    built from the trained weights produces the control the fp64 oracle computes from the same weights."""
    from conftest import load_golden
    P = load_golden("model_auv")["params"]
    plant = m.AUVModel(actionDim=6, dt=0.1, parameters=P)
    rng = np.random.default_rng(0)
    n = 6000
    x = rng.standard_normal((n, 13)) * np.array([1, 1, 1, 0, 0, 0, 0, .5, .5, .5, .2, .2, .2])
    q = rng.standard_normal((n, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
    x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    u = 300.0 * rng.standard_normal((n, 6))
    xn = plant.build_step_graph("plant", x[..., None], u[..., None])
    model = m.NNAUVModel()
    learner = m.LearnerBase(model, bufferSize=n)
    learner.add_rb(x[..., None], u[..., None], xn)
    assert learner.rb_trans()["obs"].shape == (n, 13, 1)
    learner.stats()
    assert np.all(model.Xstd > 0) and model.Xstd.shape == (16,) and model.Ystd.shape == (13,)
    before = np.mean((model.build_step_graph("nn", x[:500, :, None], u[:500, :, None]) - xn[:500]) ** 2)
    first, last = learner.train_all(learningRate=3e-3, epoch=400)
    after = np.mean((model.build_step_graph("nn", x[:500, :, None], u[:500, :, None]) - xn[:500]) ** 2)
    print("NNAUVModel trained on %d Fossen transitions: normalised loss %.4f -> %.4f, one-step MSE %.3g -> %.3g" % (n, first, last, before, after))
    assert last < 0.25 * first and after < 0.25 * before
    # the controller on the trained weights against the oracle on the same weights
    K, H = 1024, 8
    sigma = 100.0 * np.eye(6)
    goal = [1.0, 0.5, -0.5, 0, 0, 0, 1.0] + [0.0] * 6
    Q = np.array([50.0] * 3 + [5.0] * 4 + [1.0] * 6)
    h = m.Handle(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, goal=goal, Q=Q, nnauv=model.mlp(), seed=2)
    p64 = orc.Problem(tau=H, s=13, a=6, lam=1.0, sigma=sigma, goal=goal, Q=Q, nnauv=model.mlp(), threads=0, dtype=np.float64)
    p32 = orc.Problem(tau=H, s=13, a=6, lam=1.0, sigma=sigma, goal=goal, Q=Q, nnauv=model.mlp(), threads=0)
    x0 = np.array([0, 0, 0, 0, 0, 0, 1.0] + [0.0] * 6, F32)
    un = h.next(x0)
    noise = h.debug_get(m.DBG_NOISE)
    u64, U64, c64 = p64.next_with_noise(x0, np.zeros((H, 6)), noise)
    u32, _, c32 = p32.next_with_noise(x0, np.zeros((H, 6)), noise)
    assert np.isfinite(c64).all()
    c = h.debug_get(m.DBG_COSTS).astype(np.float64)
    rel = lambda a: float((np.abs(a - c64) / np.abs(c64)).max())
    e_gpu, e_cpu = np.abs(un - u64).max() / 100.0, np.abs(u32 - u64).max() / 100.0
    print("trained controller: rel cost err GPU %.3g / fp32 CPU %.3g; |du|/sigma GPU %.3g / fp32 CPU %.3g" % (rel(c), rel(c32.astype(np.float64)), e_gpu, e_cpu))
    assert rel(c) < 4 * max(rel(c32.astype(np.float64)), 1e-6)
    assert e_gpu <= max(1e-5, 4 * e_cpu)


def test_train_the_speed_network_on_the_fossen_model(m):
    """NNAUVModelSpeed (nn_model.py:307-588) through the same loop: its training pairs are (Euler angles, velocities, forces) -> the
    velocity delta; LearnerBase's statistics and Adam on the device; the trained model's one-step velocity prediction of the Fossen
    plant improves by 4x, and its pose step — the quaternion kinematics, no network involved — is the oracle's."""
    from conftest import load_golden
    P = load_golden("model_auv")["params"]
    plant = m.AUVModel(actionDim=6, dt=0.1, parameters=P)
    rng = np.random.default_rng(3)
    n = 6000
    x = rng.standard_normal((n, 13)) * np.array([1, 1, 1, 0, 0, 0, 0, .5, .5, .5, .2, .2, .2])
    q = rng.standard_normal((n, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
    x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    u = 300.0 * rng.standard_normal((n, 6))
    xn = plant.build_step_graph("plant", x[..., None], u[..., None])
    model = m.NNAUVModelSpeed(dt=0.1)
    assert [w.shape for w in model.get_weights()[0::2]] == [(15, 16), (16, 16), (16, 16), (16, 6)]
    learner = m.LearnerBase(model, bufferSize=n)
    learner.add_rb(x[..., None], u[..., None], xn)
    learner.stats()
    assert model.Xstd.shape == (15,) and model.Ystd.shape == (6,) and np.all(model.Xstd > 0)
    vel_mse = lambda: np.mean((model.build_step_graph("nn", x[:500, :, None], u[:500, :, None])[:, 7:] - xn[:500, 7:]) ** 2)
    before = vel_mse()
    first, last = learner.train_all(learningRate=3e-3, epoch=400)
    after = vel_mse()
    print("NNAUVModelSpeed trained on %d Fossen transitions: normalised loss %.4f -> %.4f, velocity MSE %.3g -> %.3g" % (n, first, last, before, after))
    assert last < 0.25 * first and after < 0.25 * before
    p64 = orc.Problem(tau=2, s=13, a=6, dt=0.1, sigma=np.eye(6), goal=np.zeros(13), nnauv_speed=model.mlp(), dtype=np.float64)
    got = model.build_step_graph("nn", x[:500, :, None], u[:500, :, None])[..., 0]
    np.testing.assert_allclose(got, p64.model_next(x[:500], u[:500]), rtol=1e-4, atol=1e-4)  # fp32 network on O(100 N) inputs against fp64


def test_weights_pushed_into_a_live_controller(m):
    """mppi_set_mlp / ControllerBase.update_model: a controller keeps running while the learner improves its model — after the push its
    step equals the step of a controller CREATED with the new weights (same seed, same step counter), for the NNAUVModel on the matrix
    cores, its split-bf16 form, the vector-ALU kernels and the point-mass 2x256 network; shapes are fixed at creation."""
    rng = np.random.default_rng(5)

    def nnauv(seed, hid=32, n_out=13, n_in=16):
        r = np.random.default_rng(seed)
        dims = [n_in, hid, hid, hid, n_out]
        return dict(W=[(r.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i]) * (0.1 if i == 3 else 1)).astype(F32) for i in range(4)],
                    b=[(r.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i]) * (0.1 if i == 3 else 1)).astype(F32) for i in range(4)],
                    xmean=r.uniform(-0.1, 0.1, n_in).astype(F32), xstd=r.uniform(0.8, 1.2, n_in).astype(F32),
                    ymean=r.uniform(-0.01, 0.01, n_out).astype(F32), ystd=r.uniform(0.8, 1.2, n_out).astype(F32))

    x13 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    base = dict(k=1024, tau=6, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=0.25 * np.eye(6), goal=[1.0, 2.0, -3.0, 0, 0, 0, 1.0] + [0.0] * 6, seed=4)
    cases = [("nnauv", dict(hid=32), {}), ("nnauv", dict(hid=32), dict(mlp_bf16x3=True)), ("nnauv", dict(hid=16), {}),
             ("nnauv_speed", dict(hid=16, n_out=6, n_in=15), {})]
    for key, shape, extra in cases:
        old, new = nnauv(1, **shape), nnauv(2, **shape)
        live = m.Handle(**base, **{key: old}, **extra)
        live.set_mlp(new)
        fresh = m.Handle(**base, **{key: new}, **extra)
        np.testing.assert_array_equal(live.next(x13), fresh.next(x13))
        with pytest.raises(m.MppiError):  # another width
            live.set_mlp(nnauv(3, **dict(shape, hid=48 - shape["hid"])))
    # point-mass 2x256 (k_rollout_mlp2): weights are re-read from memory by every launch
    def pm(seed):
        r = np.random.default_rng(seed)
        dims = [9, 256, 256, 6]
        return dict(W=[(r.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i]) * (0.1 if i == 2 else 1)).astype(F32) for i in range(3)],
                    b=[(r.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i]) * (0.1 if i == 2 else 1)).astype(F32) for i in range(3)])
    pb = dict(k=2048, tau=8, s_dim=6, a_dim=3, lam=1.0, sigma=0.25 * np.eye(3), goal=[1, 0, .5, 0, .75, 0], seed=4)
    live = m.Handle(**pb, mlp=pm(1))
    live.next(np.zeros(6, F32))
    live.set_mlp(pm(2))
    fresh = m.Handle(**pb, mlp=pm(2))
    fresh.next(np.zeros(6, F32))  # same step counter: the Philox stream depends on it
    x6 = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    # the first steps differed (other weights -> other U), so compare the rollout costs of the SAME nominal sequence instead
    U = (0.1 * rng.standard_normal((8, 3))).astype(F32)
    eps = (0.25 * rng.standard_normal((2048, 8, 3))).astype(F32)
    np.testing.assert_array_equal(live.rollout_cost(x6, U, eps), fresh.rollout_cost(x6, U, eps))
    with pytest.raises(m.MppiError):
        m.Handle(k=64, tau=4, s_dim=6, a_dim=3, sigma=np.eye(3)).set_mlp(pm(1))  # not a learned-model handle
    # the mirror classes: the learner changes the model object, update_model() carries it into the controller
    model = m.NNAUVModel(weights=nnauv(1))
    ctl = m.ControllerBase(model=model, cost=m.StaticCost(1.0, 1.0, 1.0, base["sigma"], np.array(base["goal"])[:, None], np.ones(13), diag=True),
                           k=1024, tau=6, sDim=13, aDim=6, lam=1.0, sigma=base["sigma"], seed=4)
    w2 = nnauv(2)
    model.update_weights([v for pair in zip(w2["W"], w2["b"]) for v in pair])
    ctl.update_model()
    model2 = m.NNAUVModel(weights=w2)
    ctl2 = m.ControllerBase(model=model2, cost=m.StaticCost(1.0, 1.0, 1.0, base["sigma"], np.array(base["goal"])[:, None], np.ones(13), diag=True),
                            k=1024, tau=6, sDim=13, aDim=6, lam=1.0, sigma=base["sigma"], seed=4)
    np.testing.assert_array_equal(ctl.next(x13[:, None]), ctl2.next(x13[:, None]))


def fossen_transitions(m, n, seed=0, scale=300.0):
    from conftest import load_golden
    plant = m.AUVModel(actionDim=6, dt=0.1, parameters=load_golden("model_auv")["params"])
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 13)) * np.array([1, 1, 1, 0, 0, 0, 0, .5, .5, .5, .2, .2, .2])
    q = rng.standard_normal((n, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
    x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    u = scale * rng.standard_normal((n, 6))
    return plant, x[..., None], u[..., None], plant.build_step_graph("plant", x[..., None], u[..., None])


def test_update_weights_between_trainings_reaches_the_device(m):
    """the device learner used to be built once from the model's weights and never refreshed — the reference's grid-search
    pattern model.update_weights(init_weights) (learner_base.py:143), or a loaded weight file, was ignored by the next train() and then
    overwritten by it. Now: train, put the initial weights back, train again == a fresh learner trained once from those weights."""
    _, x, u, xn = fossen_transitions(m, 2000)
    model = m.NNAUVModel()
    init = model.get_weights()
    learner = m.LearnerBase(model, bufferSize=2000)
    learner.add_rb(x, u, xn)
    learner.stats()
    learner.train_all(learningRate=3e-3, epoch=20)
    assert not np.array_equal(model.get_weights()[0], init[0])
    model.update_weights(init)
    a = learner.train_all(learningRate=3e-3, epoch=20)  # (train_all starts a new Adam, learner_base.py:149)
    fresh_model = m.NNAUVModel()
    fresh = m.LearnerBase(fresh_model, bufferSize=2000)
    fresh.add_rb(x, u, xn)
    fresh.stats()
    b = fresh.train_all(learningRate=3e-3, epoch=20)
    assert a == b
    for wa, wb in zip(model.get_weights(), fresh_model.get_weights()):
        np.testing.assert_array_equal(wa, wb)
    # evaluate() looks at the model it is given, not at what an earlier call left on the device
    X, y = model.prepare_training_data(x, xn, u)
    l_trained = learner.evaluate(model, X, y)
    model.update_weights(init)
    l_init = learner.evaluate(model, X, y)
    assert l_init > 1.15 * l_trained, (l_init, l_trained)  # 20 Adam steps: 1.09 -> 0.82


RESUME_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import mppi_tf_amd as m
path, rb, out = sys.argv[2:5]
model = m.NNAUVModel()
learner = m.LearnerBase(model, filename=rb, bufferSize=2000)
learner.load_params(path)
d = learner.rb_trans()
X, y = model.prepare_training_data(d["obs"], d["next_obs"], d["act"])
learner.train(X, y, epoch=3, learningRate=3e-3)
h = m.Handle(k=1024, tau=8, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=100.0 * np.eye(6), goal=[1.0, 0.5, -0.5, 0, 0, 0, 1.0] + [0.0] * 6,
             Q=np.array([50.0] * 3 + [5.0] * 4 + [1.0] * 6), nnauv=model.mlp(), seed=2)
u = h.next(np.array([0, 0, 0, 0, 0, 0, 1.0] + [0.0] * 6, np.float32))
np.savez(out, u=u, step=learner.step, dev_step=learner._dev.step_count(), **{"w%d" % i: w for i, w in enumerate(model.get_weights())})
"""


def test_what_the_learner_learned_survives_the_process(m, tmp_path):
    """LearnerBase.save_params (learner_base.py:66-68) / NNModel.save_params / load_params (nn_model.py:137-142): weights,
    normalisation, both Adam moments and the step count in ONE flat file (mppi_learner_save). save -> NEW PROCESS -> load -> the next Adam
    steps and the next control step are bit-identical to the ones the saving process goes on to make. The file's layout is checked
    against the documented one byte count by byte count."""
    import struct
    import subprocess
    import sys
    from conftest import ROOT
    _, x, u, xn = fossen_transitions(m, 2000, seed=4)
    model = m.NNAUVModel()
    learner = m.LearnerBase(model, bufferSize=2000, logPath=str(tmp_path / "learner"))
    learner.add_rb(x, u, xn)
    learner.stats()
    X, y = model.prepare_training_data(x, xn, u)
    learner.train(X, y, epoch=20, learningRate=3e-3)
    path = learner.save_params(learner.step)
    assert path.endswith("weights_step20")
    rb = str(tmp_path / "rb.npz")
    learner.save_rb(rb)
    # the documented layout
    raw = open(path, "rb").read()
    assert raw[:8] == b"MPPILRN1"
    n_layers, *rest = struct.unpack("<8i", raw[8:40])
    widths, step, has_norm = rest[:5], rest[5], rest[6]
    assert (n_layers, list(widths), step, has_norm) == (4, [16, 32, 32, 32, 13], 20, 1)
    n_fl = sum(3 * (widths[l] + 1) * widths[l + 1] for l in range(4))
    assert len(raw) == 40 + 4 * n_fl + 8 * 2 * (16 + 13)
    W0 = np.frombuffer(raw, "<f4", 16 * 32, 40).reshape(16, 32)
    np.testing.assert_array_equal(W0, model.get_weights()[0])
    np.testing.assert_array_equal(np.frombuffer(raw, "<f8", 16, 40 + 4 * n_fl), model.Xmean)
    # this process goes on
    learner.train(X, y, epoch=3, learningRate=3e-3)
    h = m.Handle(k=1024, tau=8, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=100.0 * np.eye(6), goal=[1.0, 0.5, -0.5, 0, 0, 0, 1.0] + [0.0] * 6,
                 Q=np.array([50.0] * 3 + [5.0] * 4 + [1.0] * 6), nnauv=model.mlp(), seed=2)
    u_here = h.next(np.array([0, 0, 0, 0, 0, 0, 1.0] + [0.0] * 6, F32))
    # a new process resumes from the file
    out = str(tmp_path / "resumed.npz")
    r = subprocess.run([sys.executable, "-c", RESUME_WORKER, ROOT, path, rb, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    assert int(d["step"]) == 23 and int(d["dev_step"]) == 23
    for i, w in enumerate(model.get_weights()):
        np.testing.assert_array_equal(d["w%d" % i], w)
    np.testing.assert_array_equal(d["u"], u_here)
    # the model's own save_params / load_params (weights + normalisation, no optimizer state): a controller built from the loaded model is the same controller
    f = model.save_params(str(tmp_path / "model"), 23)
    other = m.NNAUVModel()
    other.load_params(f)
    for wa, wb in zip(model.get_weights(), other.get_weights()):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(other.Xstd, model.Xstd)
    np.testing.assert_array_equal(other.Ymean, model.Ymean)
    with pytest.raises(m.MppiError):  # another architecture's file
        m.Learner(make_net([9, 16, 6])).load(f)
    with pytest.raises(m.MppiError):
        m.Learner.from_file(rb)  # not a learner file


def test_validate_and_the_validation_loop_of_train(m):
    """LearnerBase.validate (learner_base.py:218-320) and the every-10th-epoch validation of train (:336-358): the trajectory error of the
    model rolled from each trajectory's first state against the fp64 oracle rolling the same weights, the transition error against a
    numpy forward pass, every flag combination's return shape, and the log train(..., val=...) leaves."""
    plant, x, u, xn = fossen_transitions(m, 3000, seed=6)
    model = m.NNAUVModel()
    learner = m.LearnerBase(model, bufferSize=3000)
    learner.add_rb(x, u, xn)
    learner.stats()
    # held-out trajectories of the plant: k = 12, tau = 9
    rng = np.random.default_rng(9)
    k, tau = 12, 9
    acts = 300.0 * rng.standard_normal((k, tau, 6))
    st = np.zeros((k, 13, 1))
    st[:, 6] = 1.0
    traj = [st]
    for t in range(tau - 1):
        traj.append(plant.build_step_graph("plant", traj[-1], acts[:, t][..., None]))
    gt = np.concatenate([s[:, None] for s in traj], axis=1)[..., 0]
    X, y = model.prepare_training_data(x, xn, u)
    first, last = learner.train(X, y, epoch=35, learningRate=3e-3, val=(gt, acts))
    assert [e for e, _ in learner.val_log] == [0, 10, 20, 30] and last < first
    assert learner.val_log[-1][1]["Val-Transition-Norm"] < learner.val_log[0][1]["Val-Transition-Norm"]
    assert learner.val_log[0][1]["Val-Split"].shape == (13,) and learner.step == 35
    # the numbers, against independent evaluations of the same weights
    err, errNorm, errSplit, errSplitNorm, tErr, tErrNorm, tSplit, tSplitNorm = learner.validate(model, acts, gt, transition=True, split=True, norm=True)
    p64 = orc.Problem(tau=2, s=13, a=6, dt=0.1, sigma=np.eye(6), goal=np.zeros(13), nnauv=model.mlp(), dtype=np.float64)
    s64, rolled = gt[:, 0].copy(), [gt[:, 0].copy()]
    for t in range(tau - 1):
        s64 = p64.model_next(s64, acts[:, t])
        rolled.append(s64)
    rolled = np.stack(rolled, axis=1)
    ref_split = np.mean((rolled - gt) ** 2, axis=(0, 1))
    np.testing.assert_allclose(errSplit, ref_split, rtol=2e-3, atol=1e-9)
    assert abs(err - ref_split.mean()) <= 2e-3 * ref_split.mean()
    np.testing.assert_allclose(errSplitNorm, np.mean(((rolled - gt) / model.Ystd) ** 2, axis=(0, 1)), rtol=2e-3, atol=1e-9)
    Xn, Yn = model.prepare_training_data(gt[:, :-1].reshape(-1, 13, 1), gt[:, 1:].reshape(-1, 13, 1), acts[:, :-1].reshape(-1, 6, 1))
    hcur = Xn
    w = model.get_weights()
    for l in range(4):
        hcur = hcur @ w[2 * l].astype(np.float64) + w[2 * l + 1]
        hcur = np.maximum(hcur, 0) if l < 3 else hcur
    np.testing.assert_allclose(tSplitNorm, np.mean((hcur - Yn) ** 2, axis=0), rtol=1e-4, atol=1e-9)
    assert abs(tErrNorm - np.mean((hcur - Yn) ** 2)) <= 1e-4 * tErrNorm
    np.testing.assert_allclose(tSplit, np.mean(((hcur - Yn) * model.Ystd) ** 2, axis=0), rtol=1e-4, atol=1e-12)
    # the reference's return tuples
    assert isinstance(learner.validate(model, acts, gt), float)
    assert len(learner.validate(model, acts, gt, transition=True)) == 2 and len(learner.validate(model, acts, gt, split=True)) == 2
    assert len(learner.validate(model, acts, gt, norm=True)) == 2 and len(learner.validate(model, acts, gt, split=True, norm=True)) == 4
    assert len(learner.validate(model, acts, gt, transition=True, split=True)) == 4 and len(learner.validate(model, acts, gt, transition=True, norm=True)) == 4
