"""Census of a captured HIP graph: node types, and for every node that is not a kernel its dependencies and dependents.

The train step and the predictors replay captured graphs (trainer.TrainEngine(use_graph=True), inference.GraphedForward).
A memset NODE inside the captured step once lost its order against the kernels around it ("a hazard found
on the way"), so the step is required to consist of kernel nodes only; `census()` is what the test and
tools/graph_census.py read that from. Works on a `torch.cuda.CUDAGraph(keep_graph=True)` after capture (before or after
instantiation) through the HIP runtime's graph introspection calls - plain ctypes, no torch internals."""
from __future__ import annotations

import ctypes
from collections import Counter
from typing import Dict, List

NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "child_graph", 5: "empty", 6: "wait_event", 7: "event_record",
              8: "ext_sem_signal", 9: "ext_sem_wait", 10: "mem_alloc", 11: "mem_free", 12: "memcpy_from_symbol",
              13: "memcpy_to_symbol", 14: "batch_mem_op"}

_hip = None


def _rt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")          # already mapped by torch: the same runtime instance
    return _hip


def _check(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed with hipError {err}")


def _nodes(graph_ptr: int) -> List[int]:
    hip = _rt()
    n = ctypes.c_size_t(0)
    _check(hip.hipGraphGetNodes(ctypes.c_void_p(graph_ptr), None, ctypes.byref(n)), "hipGraphGetNodes(count)")
    arr = (ctypes.c_void_p * max(1, n.value))()
    _check(hip.hipGraphGetNodes(ctypes.c_void_p(graph_ptr), arr, ctypes.byref(n)), "hipGraphGetNodes")
    return [arr[i] for i in range(n.value)]


def _type(node: int) -> int:
    t = ctypes.c_int(-1)
    _check(_rt().hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)), "hipGraphNodeGetType")
    return t.value


def _neighbours(node: int, fn_name: str) -> List[int]:
    fn = getattr(_rt(), fn_name)
    n = ctypes.c_size_t(0)
    _check(fn(ctypes.c_void_p(node), None, ctypes.byref(n)), fn_name + "(count)")
    if n.value == 0:
        return []
    arr = (ctypes.c_void_p * n.value)()
    _check(fn(ctypes.c_void_p(node), arr, ctypes.byref(n)), fn_name)
    return [arr[i] for i in range(n.value)]


def census(graph) -> Dict:
    """graph: torch.cuda.CUDAGraph built with keep_graph=True (after capture_end), or a raw hipGraph_t address.
    -> {"nodes": N, "types": {name: count}, "max_fan_in": k, "max_fan_out": k, "roots": r,
        "non_kernel": [{"index", "type", "deps": [(index, type)], "dependents": [(index, type)]}]}"""
    ptr = graph if isinstance(graph, int) else int(graph.raw_cuda_graph())
    nodes = _nodes(ptr)
    index = {n: i for i, n in enumerate(nodes)}
    types = [_type(n) for n in nodes]
    out = {"nodes": len(nodes), "types": dict(Counter(NODE_TYPES.get(t, str(t)) for t in types)), "non_kernel": []}
    fan_in = fan_out = roots = 0
    for i, n in enumerate(nodes):
        deps = _neighbours(n, "hipGraphNodeGetDependencies")
        fan_in = max(fan_in, len(deps))
        roots += not deps
        if types[i] != 0:
            outs = _neighbours(n, "hipGraphNodeGetDependentNodes")
            out["non_kernel"].append({
                "index": i, "type": NODE_TYPES.get(types[i], str(types[i])),
                "deps": [(index.get(d, -1), NODE_TYPES.get(types[index[d]], "?") if d in index else "?") for d in deps],
                "dependents": [(index.get(d, -1), NODE_TYPES.get(types[index[d]], "?") if d in index else "?") for d in outs]})
        if i % 64 == 0 or types[i] != 0:
            fan_out = max(fan_out, len(_neighbours(n, "hipGraphNodeGetDependentNodes")))
    out.update(max_fan_in=fan_in, max_fan_out_sampled=fan_out, roots=roots)
    return out
