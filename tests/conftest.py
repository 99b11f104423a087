import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _load_package():
    """The package directory is `rust-raytracer_amd` (not an identifier), so it is loaded by path."""
    if "rust_raytracer_amd" in sys.modules:
        return sys.modules["rust_raytracer_amd"]
    path = os.path.join(ROOT, "rust-raytracer_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("rust_raytracer_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rust_raytracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


rr = _load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rrt():
    return rr


@pytest.fixture(scope="session")
def oracle_binding():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def cornell(rrt):
    scene = rrt.Scene.load_from_folder(os.path.join(rrt.SCENES, "cornell"))
    rs = rrt.load_render_settings(os.path.join(rrt.SCENES, "cornell"))
    return scene, rs


@pytest.fixture(scope="session")
def gpu_ctx(rrt):
    ctx = rrt.Context(0)
    yield ctx
    ctx.close()


# Achieved parity figures (agreement rates, error maxima, relMSE ...): tests record them here and the
# session writes them to one JSON next to the GPU run's other outputs, so that the numbers behind
# "passed" are kept and not only the thresholds.
PARITY_FIGURES = {}


@pytest.fixture(scope="session")
def parity_record():
    def record(test, **figures):
        PARITY_FIGURES.setdefault(test, {}).update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in figures.items()})
    return record


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_FIGURES:
        return
    import json
    path = os.environ.get("RRT_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(PARITY_FIGURES, f, indent=1, sort_keys=True)
    except OSError:
        pass
