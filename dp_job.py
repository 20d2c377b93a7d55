#!/usr/bin/env python
"""dp_job.py — the `--use_dp` branch of the reference's experiment driver (dp_job.sh -> Model_Code/timeautodiff_run.py
-> train_dp_autoencoder, timeautoencoder.py:603-786) as one parsing, resumable command on this engine.

Data: `--data file.npy|file.pt` holding the already windowed / encoded tensor [N, T, K] the reference calls
`processed_data` (its CSV parsing and windowing — DP.py, process_edited.py — are out of scope), or
`--synthetic N T K` for numeric-only features in [0,1).  Example (dp_job.sh settings for AirQuality):

    python dp_job.py --synthetic 9333 24 13 --hidden_size 200 --batch_size 50 --epsilon 0.1 --n_epochs 5 \\
                     --save_path experiments/dp_0.1/airquality_model.pt
"""
import argparse
import json
import sys
import time

import torch


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--data", help=".npy or .pt tensor [N,T,K] (processed_data)")
    src.add_argument("--synthetic", nargs=3, type=int, metavar=("N", "T", "K"))
    ap.add_argument("--n_bins", type=int, default=0)
    ap.add_argument("--n_cats", type=int, default=0)
    ap.add_argument("--cards", type=int, nargs="*", default=[])
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--hidden_size", type=int, default=200)
    ap.add_argument("--num_layers", type=int, default=1)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--weight_decay", type=float, default=1e-6)
    ap.add_argument("--n_epochs", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=50)
    ap.add_argument("--min_beta", type=float, default=1e-5)
    ap.add_argument("--max_beta", type=float, default=0.1)
    ap.add_argument("--emb_dim", type=int, default=128)
    ap.add_argument("--time_dim", type=int, default=8)
    ap.add_argument("--lat_dim", type=int, default=7)
    ap.add_argument("--epsilon", type=float, default=0.1)
    ap.add_argument("--delta", type=float, default=1e-5)
    ap.add_argument("--max_grad_norm", type=float, default=1.0)
    ap.add_argument("--max_physical_batch_size", type=int, default=128)
    ap.add_argument("--accountant", default="prv", choices=["prv", "rdp", "gdp"])
    ap.add_argument("--save_path")
    ap.add_argument("--load_path")
    ap.add_argument("--seed", type=int, default=None, help="fixes weights, DP noise and Poisson sampling")
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--tune_gemms", action="store_true", help="let TunableOp pick the library GEMM per shape (warm-up)")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        sys.exit("dp_job.py needs a HIP device: the DP-SGD path has no CPU fallback")
    dev = torch.device("cuda", 0)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    if args.synthetic:
        N, T, K = args.synthetic
        data = torch.rand(N, T, K, generator=torch.Generator().manual_seed(1 if args.seed is None else args.seed))
    elif args.data.endswith(".npy"):
        import numpy as np
        data = torch.from_numpy(np.load(args.data, allow_pickle=False)).float()
    else:
        data = torch.load(args.data, weights_only=True).float()
    n_nums = data.shape[2] - args.n_bins - args.n_cats
    from dp_timeautodiff_amd.train import train_dp_autoencoder
    t0 = time.time()
    info = {"n_bins": args.n_bins, "n_cats": args.n_cats, "n_nums": n_nums, "cards": args.cards}
    out = train_dp_autoencoder(
        info, data, args.channels, args.hidden_size, args.num_layers, args.lr, args.weight_decay, args.n_epochs,
        args.batch_size, 1, args.min_beta, args.max_beta, args.emb_dim, args.time_dim, args.lat_dim, dev, args.epsilon,
        args.delta, args.max_grad_norm, args.save_path, args.load_path,
        max_physical_batch_size=args.max_physical_batch_size, accountant=args.accountant,
        use_graph=not args.no_graph, tune_gemms=args.tune_gemms, noise_seed=args.seed, sampler_seed=args.seed)
    rep = getattr(train_dp_autoencoder, "last_epsilon_report", {})
    print(json.dumps({"epsilon_spent": out[7], "accountant": rep.get("accountant"),
                      "rdp_epsilon_same_ledger": rep.get("rdp_epsilon"), "delta": rep.get("delta"),
                      "target_epsilon": args.epsilon, "seconds": round(time.time() - t0, 2),
                      "samples": int(data.shape[0]), "epochs": args.n_epochs}))


if __name__ == "__main__":
    main()
