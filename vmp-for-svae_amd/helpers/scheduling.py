"""Mirror of reference helpers/scheduling.py:6-35: cartesian product of parameter lists."""
import collections.abc
from itertools import product


def create_schedule(param_ranges, verbose=False):
    lists = []
    for param, vals in param_ranges.items():
        if isinstance(vals, str) or not isinstance(vals, collections.abc.Iterable):
            vals = [vals]
        lists.append([(param, v) for v in vals])
    schedule = [dict(c) for c in product(*lists)]
    print('Created schedule containing %d configurations.' % len(schedule))
    if verbose:
        for c in schedule:
            print(c)
    return schedule
