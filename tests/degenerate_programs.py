"""Small programs at the edges of the description format (shared by the CPU and
GPU tests): operators without field accesses, scalar-only operators, 1x1xN and
1-D domains, and three malformed descriptions with the error each must raise."""

_A32 = {"a": {"data": "constant:1.0", "data_type": "float32"}}

VALID = {
    "constant_operator": {
        "inputs": dict(_A32), "outputs": ["b"], "dimensions": [4, 4, 8],
        "program": {"b": {"computation_string": "b = 1.5", "boundary_conditions": {},
                          "data_type": "float32"}}},
    "scalar_only_operator": {
        "inputs": {"s": {"data": 2.0, "data_type": "float32", "input_dims": []}},
        "outputs": ["b"], "dimensions": [4, 4, 8],
        "program": {"b": {"computation_string": "b = s * 2.0", "boundary_conditions": {},
                          "data_type": "float32"}}},
    "one_by_one_by_n": {
        "inputs": {"a": {"data": "constant:1.0", "data_type": "float64"}},
        "outputs": ["b"], "dimensions": [1, 1, 4],
        "program": {"b": {"computation_string": "b = a[i,j,k-1] + a[i+1,j,k]",
                          "boundary_conditions": {"a": {"type": "constant", "value": 2.0}},
                          "data_type": "float64"}}},
    "one_dimensional": {
        "inputs": dict(_A32), "outputs": ["b"], "dimensions": [7],
        "program": {"b": {"computation_string": "b = a[k-1] + a[k+1]",
                          "boundary_conditions": {"a": {"type": "constant", "value": 0.5}},
                          "data_type": "float32"}}},
}

INVALID = {
    "missing_boundary_condition": (ValueError, "no boundary condition", {
        "inputs": dict(_A32), "outputs": ["b"], "dimensions": [4, 4, 8],
        "program": {"b": {"computation_string": "b = a[i-1,j,k]", "boundary_conditions": {},
                          "data_type": "float32"}}}),
    "unknown_output": (RuntimeError, "not produced", {
        "inputs": dict(_A32), "outputs": ["zz"], "dimensions": [4, 4, 8],
        "program": {"b": {"computation_string": "b = a[i,j,k]",
                          "boundary_conditions": {"a": {"type": "constant", "value": 0.0}},
                          "data_type": "float32"}}}),
    "unknown_function": (ValueError, "Unsupported function", {
        "inputs": dict(_A32), "outputs": ["b"], "dimensions": [4, 4, 8],
        "program": {"b": {"computation_string": "b = frobnicate(a[i,j,k])",
                          "boundary_conditions": {"a": {"type": "constant", "value": 0.0}},
                          "data_type": "float32"}}}),
}
