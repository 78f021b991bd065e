from ... import _pkg

_m = _pkg("core.linear_assignment")
INFTY_COST = _m.INFTY_COST
min_cost_matching, matching_cascade = _m.min_cost_matching, _m.matching_cascade
gate_cost_matrix_by_mahalanobis, linear_sum_assignment = _m.gate_cost_matrix_by_mahalanobis, _m.linear_sum_assignment
