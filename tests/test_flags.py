"""far_amd/flags.py is the ONE registry of FAR_* environment switches and far_set_tuning keys (CPU-side checks; the GPU side,
tests/test_flags_gpu.py, runs each of them and holds it to its stated neutrality class)."""
import os
import re

from far_amd import flags

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources(exts, dirs):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            if '__pycache__' in base or os.sep + 'lib' in base:
                continue
            for f in files:
                if f.endswith(exts):
                    yield os.path.join(base, f)


def test_no_far_environment_read_outside_the_registry():
    bad = []
    for p in list(_sources(('.py',), ['far_amd'])) + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, 'demo.py'), os.path.join(ROOT, '__graft_entry__.py')]:
        if p.endswith(os.path.join('far_amd', 'flags.py')):
            continue
        for i, ln in enumerate(open(p), 1):
            if re.search(r"environ[^\n]*['\"]FAR_", ln) and 'FAR_COMMIT' not in ln:
                bad.append(f'{os.path.relpath(p, ROOT)}:{i}: {ln.strip()}')
    assert not bad, '\n'.join(bad)


def test_every_registered_name_is_used_and_every_used_name_is_registered():
    names = set()
    for p in list(_sources(('.py', '.sh'), ['far_amd', 'tools'])) + [os.path.join(ROOT, 'bench.py')]:
        names |= set(re.findall(r'\b(FAR_[A-Z0-9_]+)\b', open(p).read()))
    compile_time = {n for n in names if n.startswith(('FAR_WINO_', 'FAR_K9_', 'FAR_RING_', 'FAR_ONCE', 'FAR_OK', 'FAR_EVAL_CONFIG', 'FAR_LS_', 'FAR_NO_X', 'FAR_DPP', 'FAR_BUILD_ID', 'FAR_SIDE'))
                    or re.fullmatch(r'FAR_E[A-Z]+', n)}
    used = names - compile_time
    assert flags.known() <= used | {'FAR_COMMIT'}, sorted(flags.known() - used)
    assert used <= flags.known(), sorted(used - flags.known())


def test_switch_targets_exist_and_default_on():
    import far_amd.loftr  # noqa: F401
    for sw in flags.SWITCHES:
        obj, attr = flags.target(sw)
        # the product default: every feature on -- except an opt-in switch (off_value True: FAR_FPN_STREAM), which defaults off
        assert getattr(obj, attr) is (not sw.off_value) or os.environ.get(sw.env), sw.env
        assert sw.neutral in ('bitwise', 'parity') and sw.scope in ('inference', 'training')


def test_tuning_keys_match_the_library_sources():
    used = set()
    for p in _sources(('.hip', '.h', '.inc'), [os.path.join('far_amd', 'csrc')]):
        used |= {int(k) for k in re.findall(r'far_get_tuning\((\d+)\)', open(p).read())}
    assert used == {t.key for t in flags.TUNING}, (sorted(used), sorted(t.key for t in flags.TUNING))
    hdr = open(os.path.join(ROOT, 'include', 'far_hip.h')).read()
    assert 'process-global' in hdr.lower()                     # the ABI header owns up to the tuning state
