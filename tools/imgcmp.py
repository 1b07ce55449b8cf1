"""prcmp: compare two EXR images channel by channel and print the reference tool's statistics (src/tools/imgcmp/main.cpp) -- Min / Max /
Mean of both, MSE, RMSE, MAE, MAPE, PSNR, SNR, variances -- computed by the library (prgpu_image_compare).
usage: python tools/imgcmp.py INPUT.exr REFERENCE.exr [--color] [--channel NAME] [--no-global-stats] [--crop sx,sy,ex,ey | --ncrop sx,sy,ex,ey]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exr_piz import read_exr  # noqa: E402
from pearray_amd import backend  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description="Compare two images and calculate multiple statistics")
    ap.add_argument("input")
    ap.add_argument("reference")
    ap.add_argument("--color", action="store_true", help="only check the channels R, G, B")
    ap.add_argument("-c", "--channel", default="", help="check only the given channel")
    ap.add_argument("--no-global-stats", action="store_true")
    ap.add_argument("--crop", default=None, help="sx,sy,ex,ey in pixels")
    ap.add_argument("--ncrop", default=None, help="sx,sy,ex,ey normalised")
    args = ap.parse_args(argv)
    a, b = read_exr(args.input), read_exr(args.reference)
    ha, wa = next(iter(a.values())).shape
    hb, wb = next(iter(b.values())).shape
    if (ha, wa) != (hb, wb):
        print("Error: Two inputs does not match in shape", file=sys.stderr)
        return 1
    names = [n for n in a if n in b and (n == args.channel if args.channel else (not args.color or n in ("R", "G", "B")))]
    if not names:
        print("Error: Could not find the smallest common channels between the given inputs", file=sys.stderr)
        return 1
    crop = None
    if args.crop:
        crop = [int(v) for v in args.crop.split(",")]
    elif args.ncrop:
        f = [float(v) for v in args.ncrop.split(",")]
        crop = [int(wa * f[0]), int(ha * f[1]), int(wa * f[2]), int(ha * f[3])]
    stats = []
    for n in names:
        st = backend.image_compare(a[n], b[n], crop)
        stats.append(st)
        print("Channel %s>" % n)
        for label, value in backend.image_stats_report(st):
            print("  -[%-12s] = %s" % (label, value if isinstance(value, str) else "%g" % value))
    if not args.no_global_stats and len(stats) > 1:
        print("Global>")
        for label, value in backend.image_stats_report(backend.image_stats_merge(stats)):
            print("  -[%-12s] = %s" % (label, value if isinstance(value, str) else "%g" % value))
    return 0


if __name__ == "__main__":
    sys.exit(main())
