import collections.abc
import torch


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable):
        return tuple(x)
    return (x, x)


def drop_path(x, drop_prob=0., training=False):
    if not drop_prob or not training:
        return x
    keep = 1 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    mask = x.new_empty(shape).bernoulli_(keep)
    return x.div(keep) * mask
