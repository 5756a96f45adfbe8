import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus():
    """CPUs this process may really use: affinity and cgroup quota (the GPU box shows 256 CPUs with a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a -m gpu case dominated by the float64 oracle (tens of minutes of host LAPACK); runs "
                                       "only with SCLENS_TEST_SLOW=1, its log is kept under profiles/")
    # last resort against a test that never returns (every spin in the library is bounded, so this should not fire): with
    # pytest-timeout present and no limit given on the command line, a test may take 50 minutes, then the run is aborted --
    # an aborted run beats a GPU box that has to be reclaimed
    if (config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None)
            and os.environ.get("SCLENS_TEST_SLOW") != "1"):  # the slow oracle cases take longer than that by design
        config.option.timeout = 3000
        config.option.timeout_method = "thread"
    try:  # the oracle's BLAS / LAPACK calls: one thread per usable CPU, not per visible one
        from threadpoolctl import threadpool_limits

        config._sclens_blas_limit = threadpool_limits(limits=_usable_cpus())
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """the full-size cases (minutes each, order 30 000) run after everything else: a quick failure elsewhere is seen first, and
    with -x a failure in one of them does not hide the rest of the suite"""
    def late(item):
        return 1 if ("test_gpu_bench_size" in item.nodeid or "test_full_size_properties" in item.nodeid) else 0

    items.sort(key=late)  # stable: the order inside the two groups is unchanged


@pytest.fixture(scope="session")
def ctx():
    # some -m gpu tests hand torch CUDA tensors to the library (ensemble exchange): PyTorch's bundled HIP runtime has to be
    # initialised before the library's; Context() does that when torch is already imported
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from sclens_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


class _Options:
    """context options for the duration of one test: `opt(sy2sb_split_min=512)`; everything is restored at teardown"""

    def __init__(self, ctx):
        self.ctx, self.saved = ctx, {}

    def __call__(self, **kw):
        for k, v in kw.items():
            self.saved.setdefault(k, self.ctx.get_option(k))
            self.ctx.set_option(k, int(v))

    def reset(self, *names):
        for k in names:
            if k in self.saved:
                self.ctx.set_option(k, self.saved[k])

    def restore(self):
        for k, v in self.saved.items():
            self.ctx.set_option(k, v)
        self.saved.clear()


@pytest.fixture
def opt(ctx):
    o = _Options(ctx)
    yield o
    o.restore()
