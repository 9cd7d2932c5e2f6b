"""``python -m vlgp_amd FIN FOUT N_FACTORS [--max_iter N] [--min_iter N]`` -- the reference's command line
(vlgp/__main__.py:6-22): load a list of trial dicts (``.npy`` pickle or ``.npz``), fit, save the result dict."""
import argparse
import sys

from . import api, util


def cli(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vlgp_amd", description="variational Latent Gaussian Process (vLGP) on MI355X")
    ap.add_argument("fin", metavar="<path to input file>")
    ap.add_argument("fout", metavar="<path to output file>")
    ap.add_argument("n_factors", type=int, metavar="<number of factors>")
    ap.add_argument("--max_iter", type=int, default=20, help="Maximum number of iterations")
    ap.add_argument("--min_iter", type=int, default=5, help="Minimum number of iterations")
    ap.add_argument("--device", type=int, default=0, help="GPU index")
    args = ap.parse_args(argv)
    print("Loading {}".format(args.fin))
    trials = util.load(args.fin)
    if isinstance(trials, dict):  # an .npz of stacked arrays, or a saved result: take its trials
        trials = trials.get("trials", trials)
    trials = list(trials)
    print("{} loaded".format(args.fin))
    # (the reference also passes path=fout, a keyword get_config drops silently: vlgp/preprocess.py:108)
    result = api.fit(trials, args.n_factors, max_iter=args.max_iter, min_iter=args.min_iter, device=args.device)
    print("Saving {}".format(args.fout))
    util.save(result, args.fout)
    print("{} saved".format(args.fout))
    return 0


if __name__ == "__main__":
    sys.exit(cli())
