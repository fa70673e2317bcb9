"""What run_inference does about DLV_ERANGE (fp16's 65504 exceeded by a RAW, pre-normalisation activation of this checkpoint).

The reference network is fp32 (inference/sliding_window_inferer.py:205-229), so no checkpoint can overflow there.  Here the
remedy keeps fp16's 11 significant bits: every 3x3x3 conv is followed by InstanceNorm, which is invariant to a scale of its
input, so the overflowing conv block gets its 16-bit weights multiplied by 2^-k (`dlv_unet_set_conv_shift`: exact; eps scaled
to match) and the passes are repeated.  Which block: the library names the block whose INPUT overflowed (its InstanceNorm sums
are no longer finite) - the blocks feeding it are the candidates; how far: the fp32 statistics of a block stay finite when its
stored 16-bit output does not, and the library reports the largest |mean| + 8 sigma it saw per block (`dlv_range_report`).
bf16 at every level ("bf16_all": 8 significant bits, fp32's exponent range) remains the last resort.  The mixed format "bf16" keeps
fp16 at level 0 and is guarded the same way.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

from ._lib import DLV_ERANGE, DelivrHipError

MAX_SHIFT = 40
MAX_ATTEMPTS = 4
PEAK_FLOOR = 1.0  # a block is never moved blindly so far that its |mean| + 8 sigma falls below this (sigma < 2^-3: fp16 subnormals near)


def producers(layer: int) -> List[int]:
    """conv blocks whose stored raw output reaches conv block `layer` (18: the logits) through normalisation, pooling or the
    transposed conv: MONAI BasicUNet's wiring (inference/inference.py:190-197)"""
    if layer == 18:
        return [17]
    if layer in (10, 12, 14, 16):      # upcat_l.conv_0: the skip tensor of its level and the block below (through the deconv)
        level = 3 - (layer - 10) // 2
        return [2 * level + 1, layer - 1]
    return [layer - 1] if layer >= 1 else []


def next_shifts(layer: int, peaks: List[float], shifts: List[int]) -> Optional[Dict[int, int]]:
    """-> {conv block: new shift} for the blocks feeding `layer`, or None when nothing is left to try"""
    cand = producers(layer)
    hinted = {p: peaks[p] for p in cand if peaks[p] > 4096.0}
    if layer == 16 and not hinted:
        # upcat_1.conv_0 is computed as skip-half conv + folded up half P (csrc/upconv.hip): P carries the block's own weights and
        # is stored in 16 bits BEFORE the block's statistics exist, so an overflow of P shows up in block 16's own sums
        cand = [16]
    if not cand:
        return None
    out = {}
    for p in (hinted or cand):
        # bring |mean| + 8 sigma of the stored tensor to <= 1024: 64x head room above that for the tails; no hint (an overflow by
        # rare outliers of a block whose bulk is small): 6 bits at a time
        step = max(1, math.ceil(math.log2(hinted[p] / 1024.0))) if p in hinted else 6
        if p not in hinted and peaks[p] > 0.0 and peaks[p] * 2.0 ** -step < PEAK_FLOOR:
            continue  # (the library records every block's peak: this one is small - moving it would cost it its precision)
        k = min(MAX_SHIFT, shifts[p] + step)
        if k != shifts[p]:
            out[p] = k
    return out or None


def run_with_range_recovery(eng, precision: str, run: Callable[[str], None], reset: Callable[[], None],
                            log: Callable[[str], None] = print) -> str:
    """run(precision) performs the passes (raises DelivrHipError(DLV_ERANGE) on overflow); reset() zeroes what they accumulate.
    -> the precision the successful passes ran in (the one asked for, with shifted blocks, or "bf16_all" as the last resort)."""
    err = None
    base_shifts = None   # the shifts before the first recovery step (restored before the last resort)
    blind_layer = None   # layer whose last step had no hint: such a step is not repeated for the same layer
    for attempt in range(MAX_ATTEMPTS + 1):
        try:
            run(precision)
            return precision
        except DelivrHipError as e:
            if e.code != DLV_ERANGE or precision not in ("fp16", "bf16"):
                raise
            err = str(e)   # (keep the text only: the exception's traceback pins the frames - and HBM tensors - of the failed run)
        log(f"WARNING: {err}")
        layer, peaks = eng.range_report()
        shifts = eng.conv_shifts()
        plan = next_shifts(layer, peaks, shifts) if attempt < MAX_ATTEMPTS else None
        blind = not any(peaks[p] > 4096.0 for p in producers(layer))
        if plan is not None and blind and blind_layer == layer:
            plan = None  # the blind step did not move the overflow: what leaves the range is not a stored conv output
        if plan is None:
            break
        if base_shifts is None:
            base_shifts = list(shifts)
        blind_layer = layer if blind else None
        for p, k in sorted(plan.items()):
            log(f"WARNING: conv block {p}: storing its raw output scaled by 2^-{k} (InstanceNorm removes the factor) and repeating the passes in {precision}")
            eng.set_conv_shift(p, k)
        reset()
    log("WARNING: repeating the inference passes with bf16 operands at every level")
    if base_shifts is not None:  # the futile steps are taken back
        for p, k in enumerate(base_shifts):
            if eng.conv_shifts()[p] != k:
                eng.set_conv_shift(p, k)
    reset()
    run("bf16_all")
    return "bf16_all"
