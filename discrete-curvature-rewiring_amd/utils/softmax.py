"""Sampling weights for the SDRF draw — call surface of the reference's utils/softmax.py:4-10.

Stays on the host in numpy on purpose: the drawn edge must be bit-identical to the reference's
``np.random.choice(..., p=softmax(...))`` (sdrf_no_cuda.py:49-50), which pins numpy's ``exp``, its pairwise ``sum``
and the legacy MT19937 stream.  There is no max-subtraction in the reference, so none here: an overflowing
``a * tau`` yields NaNs and numpy's ``choice`` then raises ``ValueError`` exactly as it does there.
"""
import numpy as np

_INF = float('inf')


def softmax(a, tau=1):
    """``exp(tau * a)`` normalised to sum 1; for ``tau = inf`` the indicator vector of the first arg-max."""
    if tau == _INF:
        indicator = np.zeros(len(a))
        indicator[np.argmax(a)] = 1
        return indicator
    weights = np.exp(a * tau)
    return weights / weights.sum()
