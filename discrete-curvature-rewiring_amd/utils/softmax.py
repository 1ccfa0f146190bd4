"""Sampling weights for the SDRF draw — reference utils/softmax.py:4-10.

Stays on the host in numpy on purpose: the drawn edge must be bit-identical to
the reference's ``np.random.choice(..., p=softmax(...))`` (sdrf_no_cuda.py:49-50),
which pins numpy's ``exp``, pairwise ``sum`` and the legacy MT19937 stream.
No max-subtraction: an overflowing ``a * tau`` gives NaNs and numpy's ``choice``
raises ``ValueError``, as it does in the reference.
"""
import numpy as np


def softmax(a, tau=1):
    if tau == float('inf'):
        r = np.zeros(len(a))
        r[np.argmax(a)] = 1
        return r
    exp_a = np.exp(a * tau)
    return exp_a / exp_a.sum()
