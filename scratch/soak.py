"""One-off differential soak: run the seed-driven GPU parity tests of tests/ with shifted random seeds.
usage: python scratch/soak.py <n_offsets> [first_offset]   (progress on stdout; exits non-zero at the first mismatch)"""
import importlib, inspect, itertools, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from object_detectors_amd import hip_ops

TARGETS = {
    "test_gpu_parity": ["test_preprocess_bit_exact", "test_letterbox_bit_exact_uniform", "test_letterbox_ragged_batch_fp16_stretch_and_rgb",
                        "test_nms_synthetic_heads", "test_nms_dense_scene_high_classes_and_max_det", "test_nms_odd_anchor_count_and_many_candidates",
                        "test_finalize_scale_clip_threshold", "test_process_mask_iou", "test_postprocess_chain_seg"],
    "test_gpu_obb": ["test_rotated_nms_and_finalize", "test_rotated_nms_more_candidates_than_the_lds_sort_buffer"],
    "test_gpu_paste": ["test_paste_masks_vs_torch_grid_sample"],
    "test_gpu_pose": ["test_scale_coords_kernel"],
    "test_gpu_revert": ["test_revert_points_golden_and_random", "test_revert_masks_letterbox_undo_1080p",
                        "test_revert_masks_rows_that_are_not_whole_lines"],
    "test_gpu_nms_pairs": ["test_lists_equal_the_sweep_and_the_oracle_over_batch_sizes", "test_lists_random_settings_ties_caps_and_dtypes"],
}

def param_sets(fn):
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    axes = []
    for m in marks:
        names = [n.strip() for n in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
        vals = []
        for v in m.args[1]:
            v = getattr(v, "values", v)
            vals.append(dict(zip(names, v if len(names) > 1 else [v])))
        axes.append(vals)
    for combo in itertools.product(*axes) if axes else [()]:
        kw = {}
        for d in combo:
            kw.update(d)
        yield kw

real_rng = np.random.default_rng
n_off = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ran = 0
skipped = 0
t0 = time.time()
for off in range(first, first + n_off):
    np.random.default_rng = lambda seed=None, _o=off: real_rng(seed + 7919 * _o if isinstance(seed, (int, np.integer)) else seed)
    for mod, names in TARGETS.items():
        M = importlib.import_module(mod)
        for name in names:
            fn = getattr(M, name)
            for kw in param_sets(fn):
                if "ops" in inspect.signature(fn).parameters:
                    kw["ops"] = hip_ops
                try:
                    fn(**kw)
                except AssertionError as e:
                    # test preconditions on the generated scene (decision margin of the rotated NMS, non-empty reference
                    # masks / detections), not mismatches: the committed seeds satisfy them, shifted ones need not
                    line = traceback.extract_tb(e.__traceback__)[-1].line or ""
                    if "pick another" in str(e) or any(k in line for k in ("ref.sum() > 0", "len(ra) == 300", "len(boxes) > 0", "margin >")):
                        skipped += 1
                        continue
                    np.random.default_rng = real_rng
                    print(f"FAILED offset {off}: {mod}.{name} {dict((k, v) for k, v in kw.items() if k != 'ops')}", flush=True)
                    traceback.print_exc()
                    sys.exit(1)
                except Exception:
                    np.random.default_rng = real_rng
                    print(f"FAILED offset {off}: {mod}.{name} {dict((k, v) for k, v in kw.items() if k != 'ops')}", flush=True)
                    traceback.print_exc()
                    sys.exit(1)
                ran += 1
    torch.cuda.synchronize()
    print(f"offset {off}: {ran} cases so far ({skipped} skipped on a test precondition), {time.time() - t0:.0f}s", flush=True)
np.random.default_rng = real_rng
print("soak ok:", ran, "cases")
