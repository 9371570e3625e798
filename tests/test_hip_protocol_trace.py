"""GPU: INTEGRATION.md option A, dynamically.  The reference's own `BreakpointModel.fit` / `optimal_cn` / `get_model_data`
(remixt/cn_model.py:286-297, 354-428, 482-604) were run in the build container over a recording proxy of the reference kernel
object (oracle/make_protocol_trace.py); the fixtures hold, in order, every constructor call, attribute write, attribute read and
method call the reference host class made, with the values the reference kernel answered.  Here the same sequence is made call by call
on `remixt_amd.bpmodel.RemixtModel`: same arguments in, the reference's answers expected out (1e-6 relative -- the requirement --
on floating point, exact on integers and on `infer_cn`)."""
import pytest

from .protocol_replay import replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', ['m2', 'm3_nonormal'])
def test_reference_fit_call_trace_replays_on_the_hip_kernel_object(case):
    from remixt_amd import bpmodel
    replay(bpmodel.RemixtModel, case)
