"""The example programs run as the reference's do: execute, look at the verdict."""
import importlib.util
import os

import pytest

from wavegen import ROOT

pytestmark = pytest.mark.gpu


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("num_backends", [4, 16, 65536])
def test_integer_addition(num_backends):
    assert load("integer_addition").main(num_backends, 4, verbose=False)
