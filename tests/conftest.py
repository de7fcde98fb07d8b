import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _host_threads():
    """Host cores this process may really use (bench.host_threads): a GPU box hands out one GPU's share of the machine through a
    cgroup quota, while torch sizes its CPU thread pool by the whole machine -- 256 threads on 16 cores make the oracle's convs crawl."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


@pytest.fixture(scope='session', autouse=True)
def _cpu_threads():
    import torch
    torch.set_num_threads(_host_threads())
    yield


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


GOLDEN = os.path.join(ROOT, 'tests', 'golden')
