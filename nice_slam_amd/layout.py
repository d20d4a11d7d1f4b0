"""Parameter-blob layout shared by the host code and the kernels (csrc/nsr_layout.h).

One flat fp32 blob per decoder, tensors concatenated in the order of the reference module's
``named_parameters()`` (src/conv_onet/models/decoder.py:124-159 for MLP, :235-245 for MLP_no_xyz).
"""
from __future__ import annotations

from typing import List, Tuple

C_DIM = 32
HIDDEN = 32
EMB = 93


def param_spec(slot: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """[(name relative to '<slot>_decoder.', shape)] in blob order."""
    if slot == "coarse":
        spec = []
        for i in range(5):
            spec += [(f"pts_linears.{i}.weight", (HIDDEN, HIDDEN + (C_DIM if i == 3 else 0))),
                     (f"pts_linears.{i}.bias", (HIDDEN,))]
        spec += [("output_linear.weight", (1, HIDDEN)), ("output_linear.bias", (1,))]
        return spec
    cd = 2 * C_DIM if slot == "fine" else C_DIM
    nout = 4 if slot == "color" else 1
    spec = []
    for i in range(5):
        spec += [(f"fc_c.{i}.weight", (HIDDEN, cd)), (f"fc_c.{i}.bias", (HIDDEN,))]
    spec += [("embedder._B", (3, EMB))]
    ins = (EMB, HIDDEN, HIDDEN, EMB + HIDDEN, HIDDEN)
    for i in range(5):
        spec += [(f"pts_linears.{i}.weight", (HIDDEN, ins[i])), (f"pts_linears.{i}.bias", (HIDDEN,))]
    spec += [("output_linear.weight", (nout, HIDDEN)), ("output_linear.bias", (nout,))]
    return spec


def param_count(slot: str) -> int:
    n = 0
    for _, shp in param_spec(slot):
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


def stage_slots(stage: str) -> Tuple[str, ...]:
    """Decoders (= grids) NICE.forward touches in a stage (decoder.py:317-342)."""
    return {"coarse": ("coarse",), "middle": ("middle",), "fine": ("middle", "fine"),
            "color": ("middle", "fine", "color")}[stage]
