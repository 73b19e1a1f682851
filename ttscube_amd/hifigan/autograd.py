"""Training path of the Generator (placeholder until the conv dgrad/wgrad kernels land)."""


def generator_forward_with_grad(gen, x):
    raise NotImplementedError('Generator training (autograd) path is not built yet; wrap inference in torch.no_grad()')
