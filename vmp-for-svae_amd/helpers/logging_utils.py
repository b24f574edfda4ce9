"""Mirror of reference helpers/logging_utils.py:12-44 (run identifier from a config)."""
import numpy as np


def generate_log_id(config, method_key='method', dataset_key='dataset'):
    log_id = '%s_%s' % (config.get(method_key, 'unknownM'), config.get(dataset_key, 'unknownD'))
    for key, val in sorted(config.items()):
        if key in (method_key, dataset_key):
            continue
        if isinstance(val, str):
            txt = val
        elif isinstance(val, (int, np.integer)):
            txt = '%d' % val
        elif isinstance(val, float):
            txt = '%.5f' % val if np.log10(np.abs(val)) >= -5 else ('%.20f' % val).rstrip('0')
        else:
            raise NotImplementedError
        log_id += '_%s%s' % (key, txt)
    return log_id
