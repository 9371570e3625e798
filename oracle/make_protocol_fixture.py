"""Writes tests/golden/protocol_names.json: the NAMES the reference's host class touches on its kernel object and the names
the reference kernel class declares -- read from the reference's source as text (no import), build container only.

  cn_model_uses   every `self.model.<name>` in remixt/cn_model.py (what BreakpointModel needs from RemixtModel)
  pipeline_uses   every `model.<name>` in remixt/analysis/pipeline.py fit() (what the restart driver needs from BreakpointModel)
  pyx_public      the `cdef public` attributes of RemixtModel (remixt/bpmodel.pyx:398-456)
  pyx_cpdef       its `cpdef` methods

tests/test_protocol_static.py checks remixt_amd.bpmodel.RemixtModel / remixt_amd.cn_model.BreakpointModel against the
lists (INTEGRATION.md option A cannot be executed anywhere -- the reference never reaches a GPU -- so the static check is the
test it gets), and, where /root/reference is mounted, that the lists are current."""
import json
import os
import re
import sys

REF = os.environ.get('REMIXT_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'protocol_names.json')


def extract(ref=REF):
    cn = open(os.path.join(ref, 'remixt', 'cn_model.py')).read()
    pyx = open(os.path.join(ref, 'remixt', 'bpmodel.pyx')).read()
    pipe = open(os.path.join(ref, 'remixt', 'analysis', 'pipeline.py')).read()
    uses = sorted(set(re.findall(r'self\.model\.([A-Za-z_][A-Za-z_0-9]*)', cn)))
    fit = pipe[pipe.index('def fit('):]
    fit = fit[:fit.index('\ndef ', 1)]
    pipeline_uses = sorted(set(re.findall(r'(?<![A-Za-z_0-9.])model\.([A-Za-z_][A-Za-z_0-9]*)', fit)))
    public = sorted(set(re.findall(r'^\s*cdef public [^\n]*?([A-Za-z_][A-Za-z_0-9]*)\s*$', pyx, flags=re.M)))
    cpdef = sorted(set(re.findall(r'^\s*cpdef [^\n(]*?([A-Za-z_][A-Za-z_0-9]*)\s*\(', pyx, flags=re.M)))
    return {'cn_model_uses': uses, 'pipeline_uses': pipeline_uses, 'pyx_public': public, 'pyx_cpdef': cpdef}


if __name__ == '__main__':
    data = extract()
    json.dump(data, open(OUT, 'w'), indent=1, sort_keys=True)
    print('wrote', OUT, dict((k, len(v)) for k, v in data.items()))
