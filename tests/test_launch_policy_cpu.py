"""CPU: the launch policy state machine (slnlp/launch.py) -- graph capture first, eager warm-up, PROBE timed steps of
each mode, then the faster one for good.  torch.cuda events are replaced by a fake clock."""
import pytest

from slnlp import launch


class FakeEvent:
    now = 0.0

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        self.t = FakeEvent.now

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return other.t - self.t


@pytest.mark.parametrize("graph_ms,eager_ms,expect", [(3.5, 3.2, "eager"), (8.9, 9.4, "graph")])
def test_policy_probes_then_sticks(monkeypatch, graph_ms, eager_ms, expect):
    monkeypatch.setattr(launch.torch.cuda, "Event", FakeEvent)
    FakeEvent.now = 0.0
    calls = []

    def run(mode, ms):
        def f():
            calls.append(mode)
            FakeEvent.now += ms
            return mode
        return f

    pol = launch.LaunchPolicy()
    n = 2 + 2 * launch.PROBE + 5
    out = [pol.run("k", run("graph", graph_ms), run("eager", eager_ms)) for _ in range(n)]
    assert calls[0] == "graph" and calls[1] == "eager"                                    # capture, then warm-up
    assert calls[2:2 + launch.PROBE] == ["graph"] * launch.PROBE
    assert calls[2 + launch.PROBE:2 + 2 * launch.PROBE] == ["eager"] * launch.PROBE
    assert pol.mode("k") == expect and out[-5:] == [expect] * 5
    assert pol.mode("other batch size") is None                                           # state is per key
