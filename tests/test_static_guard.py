"""Host logic of dgl_amd.static_features (no GPU): the announcement is checked against the tensor's
in-place version counter, address, shape and strides at every use (VERDICT r5 Next #7)."""
import torch

from dgl_amd import sparse_kernels as sk


def test_token_survives_reads_and_dies_on_in_place_writes():
    t = torch.rand(10, 4)
    sk.static_features(t)
    tok = sk._static_token(t)
    assert tok != 0 and sk._static_token(t) == tok          # reading does not change anything
    _ = t + 1, t.sum(), t[2:4]
    assert sk._static_token(t) == tok
    t.mul_(2)
    assert sk._static_token(t) == 0 and id(t) not in sk._static
    sk.static_features(t)
    assert sk._static_token(t) not in (0, tok)              # a fresh token: the library re-makes its copy
    t[0, 0] = 5.0                                           # indexed assignment bumps the counter too
    assert sk._static_token(t) == 0


def test_optimizer_step_and_set_withdraw_the_promise():
    w = torch.rand(8, 1, requires_grad=True)
    sk.static_features(w)
    assert sk._static_token(w) != 0
    (w * 2).sum().backward()
    assert sk._static_token(w) != 0                         # backward does not write w
    torch.optim.SGD([w], lr=0.1).step()
    assert sk._static_token(w) == 0
    v = torch.rand(8, 1)
    sk.static_features(v)
    v.set_(torch.rand(8, 1))                                # same object, other storage
    assert sk._static_token(v) == 0


def test_other_tensors_and_release():
    a, b = torch.rand(3), torch.rand(3)
    sk.static_features(a)
    assert sk._static_token(b) == 0
    sk.release_static(a)
    assert sk._static_token(a) == 0
    sk.static_features(a)
    key = id(a)
    del a
    assert key not in sk._static                            # the weakref callback removes the entry
