"""Achieved-error log of the GPU parity tests.  Every relerr()/abserr() evaluation of the `-m gpu` test modules is
recorded here under the id of the running test; at session end the maxima are written to
gpurun_out/r03_parity_errors.json (merged back from the GPU box; the copy to be judged is committed under profiles/)."""
import json
import os

_LOG = {}


def record(kind, err, tol=None, what=None):
    test = os.environ.get('PYTEST_CURRENT_TEST', 'unknown').split(' ')[0]
    test = test.split('::', 1)[-1] if '::' in test else test
    key = kind if what is None else '%s:%s' % (kind, what)
    e = _LOG.setdefault(test, {}).setdefault(key, {'max_err': 0.0, 'n': 0})
    err = float(err)
    if not (err <= e['max_err']):          # keeps NaN visible
        e['max_err'] = err
    e['n'] += 1
    if tol is not None:
        e['tol'] = float(tol) if 'tol' not in e else max(e['tol'], float(tol))
    return err


def dump(root):
    if not _LOG:
        return None
    out_dir = os.path.join(root, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, 'r03_parity_errors.json')
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(_LOG)
    with open(path, 'w') as f:
        json.dump(old, f, indent=1, sort_keys=True)
    return path
