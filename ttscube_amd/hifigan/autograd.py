"""Training path of the Generator: differentiable forward on torch-ROCm ops (see networks/training.py)."""


def generator_forward_with_grad(gen, x):
    from ..networks.training import generator_forward_train
    return generator_forward_train(gen, x)
