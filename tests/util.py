"""Helpers shared by the parity tests."""
import re

_NEG_ZERO = re.compile(r"(?<=;)-0\.0(?=[\t\n]|$)")


def neg_zero(text: str) -> str:
    """round(1 - betabinom.cdf(...), 4) prints `-0.0` when scipy's sum lands a hair above 1; the device's tail
    prints `0.0`.  Only a WHOLE p-value token — the third ';'-field of Rest_BC / Rest_CC — is rewritten: a value such as -0.05, or
    `-0.0` inside any other field, stays as it is and would fail the comparison."""
    return _NEG_ZERO.sub("0.0", text)
