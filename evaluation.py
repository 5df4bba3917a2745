#!/usr/bin/env python3
"""Drop-in evaluation driver (reference evaluation.py:269-314): same flags, same input files, same JSON outputs -- with the
metrics computed by the HIP kernels of t2ms_amd.metrics (csrc/t2s_eval.hip) instead of numpy / scipy / dtaidistance loops.

    python evaluation.py --dataset_name ETTh1_96 --cfg_scale 9.0 --total_step 10 [--method_list MSE,WAPE,MRR,CRPS,C-FID,ED,DTW]

Kept from the reference: the path derivations (`{save_path}/generation/{backbone}_{denoiser}_{dataset}_{cfg}_{steps}/`, its
`run_0 .. run_9/`, results under `{save_path}/evaluation/{model_name}/{model_name}_{dataset}_{time}[_multi].json`), the
files each metric reads (evaluation.py:285-314: `x_1` of run_0 against the base directory's `x_t` for MSE / WAPE / C-FID,
`x_1` of the last run against the ten `x_t` stacked on a trailing axis for MRR / CRPS), the result keys, the print lines.
Additions: `ED` and `DTW` may be named in --method_list (the reference defines both, evaluation.py:137-163, and never
calls them); `--align_runs` re-orders every run's rows to run_0's ground-truth order first -- each `infer()` call shuffles
its test loader independently (dataloader.py:111; the reference never seeds), so WITHOUT it row i of one file is not row i of
another, here exactly as in the reference.
"""
import argparse
import datetime
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from t2ms_amd import metrics as M                              # noqa: E402


def _methods(method_list):
    if isinstance(method_list, list):
        return method_list
    return [m.strip() for m in method_list.strip("[]").split(",")]


def _divider():
    print("=" * 60)


def _save(result, args, suffix):
    stamp = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    path = os.path.join(args.evaluation_save_path, f"{args.model_name}_{args.dataset_name}_{stamp}{suffix}.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(result, f, indent=4)
    print(f"Evaluation denoiser_results saved to {path}.")
    return path


def evaluate_data(args, ori_data, gen_data):
    """evaluation.py:210-266 on (N, 1, L) arrays: C-FID, MSE, WAPE (+ ED, DTW) -> result dict, written as JSON."""
    _divider()
    print(f"Evalution with settings:{args}")
    methods = _methods(args.method_list)
    if gen_data is None:
        print("Error: Generated data not found.")
        return None
    if ori_data.shape != gen_data.shape:
        print(f"Original data shape: {ori_data.shape}, Generated data shape: {gen_data.shape}.")
        print("Error: Generated data does not have the same shape with original data.")
        return None
    ori = np.transpose(ori_data, (0, 2, 1)).astype(np.float32)    # (N, L, 1): the layout of the .npy files
    gen = np.transpose(gen_data, (0, 2, 1)).astype(np.float32)
    result = {}
    if "C-FID" in methods:
        from t2ms_amd.ts2vec import initialize_ts2vec            # evaluation.py:238-243: TS2Vec trained on the originals
        model = initialize_ts2vec(ori, device=args.device)
        result["C-FID"] = M.fid(model.encode(ori, encoding_window="full_series"), model.encode(gen, encoding_window="full_series"))
    if "MSE" in methods or "WAPE" in methods:
        mse, wape, _ = M.mse_wape(ori, gen, device=args.device)
        if "MSE" in methods:
            result["MSE"] = mse
        if "WAPE" in methods:
            result["WAPE"] = wape
    if "ED" in methods:
        result["ED"] = M.ed(ori, gen, device=args.device)[0]
    if "DTW" in methods:
        result["DTW"] = M.dtw(ori, gen, device=args.device)[0]
    _save(result, args, "")
    print(f"Evaluation done. Results:{result}.")
    _divider()
    return result


def evaluate_muldata(args, ori_data, gen_data):
    """evaluation.py:87-124 on ori (N, L, 1) and gen (N, L, 1, runs): CRPS, MRR."""
    _divider()
    print(f"Evalution with settings:{args}")
    methods = _methods(args.method_list)
    if gen_data is None:
        print("Error: Generated data not found.")
        return None
    result = {}
    if "CRPS" in methods:
        result["CRPS"] = M.crps(ori_data, gen_data, device=args.device)[0]
    if "MRR" in methods:
        result["MRR"] = M.mrr(ori_data, gen_data, device=args.device)[0]
    _save(result, args, "_multi")
    print(f"Evaluation done. Results:{result}.")
    _divider()
    return result


def _row_order(x_1):
    """A canonical order of the ground-truth rows (every run holds the same rows, each in its own shuffle)."""
    return np.lexsort(x_1.reshape(x_1.shape[0], -1).T)


def build_parser():
    p = argparse.ArgumentParser(description="Train flow matching model")
    p.add_argument("--method_list", type=str, default="MSE,WAPE,MRR", help="metric list [MSE,WAPE,MRR]")
    p.add_argument("--save_path", type=str, default="./results/denoiser_results", help="Denoiser Model save path")
    p.add_argument("--dataset_name", type=str, default="ETTh1_96", help="dataset name")
    p.add_argument("--backbone", type=str, default="flowmatching", help="flowmatching or DDPM or EDM")
    p.add_argument("--denoiser", type=str, default="DiT", help="DiT or MLP")
    p.add_argument("--cfg_scale", type=float, default=9.0, help="CFG Scale")
    p.add_argument("--total_step", type=int, default=10, help="total step sampled from [0,1]")
    p.add_argument("--align_runs", action="store_true",
                   help="re-order every run's rows to one ground-truth order before comparing (see the module docstring)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        sys.exit("evaluation.py: no GPU visible -- the metrics kernels run on the GPU (no CPU fallback)")
    args.device = "cuda"
    args.data_length = args.dataset_name.split("_")[-1] if args.dataset_name != "SUSHI" else 2048
    args.model_name = "{}_{}_{}_{}_{}".format(args.backbone, args.denoiser, args.dataset_name, args.cfg_scale, args.total_step)
    args.generation_save_path = os.path.join(args.save_path, "generation", args.model_name)
    args.evaluation_save_path = os.path.join(args.save_path, "evaluation", args.model_name)
    g = args.generation_save_path
    x_1 = np.load(os.path.join(g, "run_0", "x_1.npy"))                       # evaluation.py:285-286
    x_t = np.load(os.path.join(g, "x_t.npy"))
    if args.align_runs and os.path.exists(os.path.join(g, "x_1.npy")):
        base = np.load(os.path.join(g, "x_1.npy"))
        x_1, x_t = base[_row_order(base)], x_t[_row_order(base)]
    single = evaluate_data(args, ori_data=np.transpose(x_1, (0, 2, 1)), gen_data=np.transpose(x_t, (0, 2, 1)))
    runs = []
    for run_index in range(10):                                                # evaluation.py:301-313
        d = os.path.join(g, f"run_{run_index}")
        x_1 = np.load(os.path.join(d, "x_1.npy"))
        x_r = np.load(os.path.join(d, "x_t.npy"))
        if args.align_runs:
            o = _row_order(x_1)
            x_1, x_r = x_1[o], x_r[o]
        runs.append(np.expand_dims(x_r, axis=-1))
    multi = evaluate_muldata(args, ori_data=x_1, gen_data=np.concatenate(runs, axis=-1))
    return single, multi


if __name__ == "__main__":
    main()
