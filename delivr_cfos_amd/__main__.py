"""``python -m delivr_cfos_amd [config.json]`` - the reference's step-level CLI (__main__.py:47-240)
for the steps this package accelerates: BLOB_DETECTION (sliding-window inference) and
POSTPROCESSING (connected components + cell table).  Same config keys, same folder layout, same
``HOOK:`` progress lines on stdout (consumed by the Fiji plugin).  MASK_DOWNSAMPLE runs when the raw
planes are readable without an external codec (see downsample.downsample_and_mask); the atlas /
region / visualisation steps wrap closed external binaries and are out of scope: their flags are
reported and skipped.
"""
from __future__ import annotations

import argparse
import json
import os
import sys


def setup_config(settings: dict) -> dict:
    """reference __main__.py:36-44: with FLAGS.ABSPATHS false every *input*/*output*/*collection*
    location below the top level is prefixed with the top-level output_location."""
    if not settings["FLAGS"]["ABSPATHS"]:
        base = settings["output_location"]
        for section, block in settings.items():
            if not isinstance(block, dict) or section == "FLAGS":
                continue
            for key, val in block.items():
                if isinstance(val, str) and any(t in key for t in ("input", "output", "collection")):
                    block[key] = os.path.join(base, val.lstrip("/")) if not val.startswith(base) else val
    return settings


def setup_folders(settings: dict) -> None:
    """reference __main__.py:17-34"""
    for section in ("mask_detection", "blob_detection", "postprocessing"):
        loc = settings.get(section, {}).get("output_location")
        if loc:
            os.makedirs(loc, exist_ok=True)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="delivr_cfos_amd")
    ap.add_argument("config", nargs="?", default="config.json")
    ns = ap.parse_args(argv)
    if not os.path.isfile(ns.config):
        print(f"config {ns.config} not found", file=sys.stderr)
        return 2
    with open(ns.config) as fh:
        settings = json.load(fh)
    settings = setup_config(settings)
    setup_folders(settings)
    flags = settings["FLAGS"]
    steps = [k for k in ("MASK_DOWNSAMPLE", "BLOB_DETECTION", "POSTPROCESSING", "ATLAS_ALIGNMENT", "REGION_ASSIGNMENT",
                         "VISUALIZATION") if flags.get(k)]
    print(f"HOOK:OVERALL:{len(steps)}")
    step_no = 0
    from .downsample.downsample_and_mask import downsample_mask, get_real_size

    if flags.get("MASK_DOWNSAMPLE"):
        step_no += 1
        brains = sorted(os.listdir(settings["raw_location"]))
        for i, brain in enumerate(brains):
            print(f"HOOK:{step_no}:{len(steps)}:{i}:{len(brains)}")
            # (reference __main__.py:98: the step is done when masked_niftis exists - the brain's folder alone may hold only
            # ilastik's output)
            if os.path.exists(os.path.join(settings["mask_detection"]["output_location"], brain, "masked_niftis")):
                print(f"{brain} exists, skipping...")
                continue
            downsample_mask(settings, brain)
    if flags.get("BLOB_DETECTION"):
        step_no += 1
        from .inference.inference import run_inference

        from . import hostio

        mask_out = settings["blob_detection"]["input_location"]
        brains = sorted(os.listdir(mask_out))

        def niftis_of(brain):
            nifti_dir = os.path.join(mask_out, brain, "masked_niftis")
            return sorted(os.path.join(nifti_dir, f) for f in os.listdir(nifti_dir) if f.endswith(".npy"))

        # the brains are pipelined: while one runs its passes the next one's volume is read into HBM and the previous one's
        # binaries.npy streams out (run_inference: prefetch / defer_write) - per brain the step costs its passes
        try:
            for i, brain in enumerate(brains):
                print(f"HOOK:{step_no}:{len(steps)}:{i}:{len(brains)}")
                stack_shape = (1, 1, *get_real_size(os.path.join(settings["raw_location"], brain)))
                nxt = niftis_of(brains[i + 1])[:1] if i + 1 < len(brains) else []
                run_inference(niftis=niftis_of(brain), output_folder=settings["blob_detection"]["output_location"],
                              stack_shape=stack_shape, model_weights=settings["blob_detection"]["model_location"],
                              tta=flags["TEST_TIME_AUGMENTATION"], comment=brain, load_all_ram=flags["LOAD_ALL_RAM"],
                              settings=settings, prefetch=nxt[0] if nxt else None, defer_write=True)
        finally:
            hostio.wait_deferred()  # every binaries.npy is complete before the next step reads it
    if flags.get("POSTPROCESSING"):
        step_no += 1
        from .count_blobs import count_blobs

        from . import hostio

        path_in = settings["postprocessing"]["input_location"]
        brains = sorted(os.listdir(path_in))
        try:
            for i, brain in enumerate(brains):
                print(f"HOOK:{step_no}:{len(steps)}:{i}:{len(brains)}")
                stack_shape = (1, 1, *get_real_size(os.path.join(settings["raw_location"], brain)))
                count_blobs(settings, path_in, i, brain, stack_shape, settings["postprocessing"]["min_size"],
                            settings["postprocessing"]["max_size"], defer_write=True)  # (the label file streams out behind the next brain)
        finally:
            hostio.wait_deferred()
    for k in ("ATLAS_ALIGNMENT", "REGION_ASSIGNMENT"):
        if flags.get(k):
            step_no += 1
            print(f"HOOK:{step_no}:{len(steps)}:0:0")
            print(f"{k}: wraps external binaries / table work - not part of the accelerated path, skipped")
    if flags.get("VISUALIZATION"):
        # reference __main__.py:210-221; needs the cells_<brain>.csv written by the (external) region assignment step
        step_no += 1
        from .blob_highlighter import blob_highlighter

        print("Visualization")
        brains = sorted(os.listdir(settings["visualization"]["input_prediction_location"]))
        for i, brain in enumerate(brains):
            print(f"HOOK:{step_no}:{len(steps)}:{i}:{len(brains)}")
            csv_dir = settings["visualization"]["input_csv_location"]
            if not os.path.isdir(csv_dir) or not any("cells_" + brain in x for x in os.listdir(csv_dir)):
                print(f"VISUALIZATION: no cells_{brain}*.csv under {csv_dir} (region assignment did not run) - skipped")
                continue
            stack_shape = (1, 1, *get_real_size(os.path.join(settings["raw_location"], brain)))
            blob_highlighter(settings, [brain, ""], stack_shape)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
