"""pytest plugin of tools/run_reference_tests.py: keeps THIS repository's ``birdnet_stm32`` package in front of the reference's.

The reference's ``tests/conftest.py`` puts the reference's root at ``sys.path[0]`` when it is loaded; before every test module is imported this
plugin moves ``birdnet-stm32_amd/`` back to the front and forgets any ``birdnet_stm32`` module that was imported from elsewhere."""
import os
import sys

OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "birdnet-stm32_amd")


def _front():
    if sys.path[:1] != [OURS]:
        while OURS in sys.path:
            sys.path.remove(OURS)
        sys.path.insert(0, OURS)
    for name, mod in list(sys.modules.items()):
        if (name == "birdnet_stm32" or name.startswith("birdnet_stm32.")) and not (getattr(mod, "__file__", None) or "").startswith(OURS):
            del sys.modules[name]


def pytest_collect_file(file_path, parent):
    _front()
    return None


def pytest_runtest_setup(item):
    _front()
