"""The two numpy helpers of the reference's top-level utils.py that the hot path uses."""


def find_floor_in_list(keys, value):
    """Largest key <= value (learning-rate schedule lookup, reference utils.py:70-84) -> (key, index)."""
    best, best_i = None, None
    for i, k in enumerate(keys):
        if k <= value and (best is None or k > best):
            best, best_i = k, i
    if best is None:
        raise ValueError("no schedule key <= %s" % (value,))
    return best, best_i


def list_mean(lst):
    return sum(lst) / float(len(lst))
