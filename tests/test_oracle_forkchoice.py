"""CPU: the L0 literal oracle against the known answers the reference's prose pins (K1..K10) and the upstream
scenario names.  The same scenarios run against the engine in tests/test_gpu_forkchoice.py."""
import pytest

from tests import fc_scenarios
from tests.scenario import new_world


@pytest.mark.parametrize("scenario", fc_scenarios.ALL, ids=lambda f: f.__name__)
def test_scenario_on_literal_oracle(scenario):
    scenario(lambda n, **kw: new_world(n, "minimal", **kw))
