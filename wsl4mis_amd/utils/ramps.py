"""Consistency-weight ramps (ref: utils/ramps.py:19-26 sigmoid_rampup; host-side scalar arithmetic)."""
import math


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t)^2), t = clip(current / rampup_length, 0, 1); 1.0 when rampup_length == 0."""
    if rampup_length == 0:
        return 1.0
    t = min(max(float(current), 0.0), float(rampup_length)) / rampup_length
    return float(math.exp(-5.0 * (1.0 - t) ** 2))
