"""bench.py modes beside the headline: computeLD (configs[1]), jepegmix (configs[4]), e2e (files -> table)."""


def run_computeld(args, rig):
    raise SystemExit("--mode computeLD: not built yet")


def run_jepegmix(args, rig):
    raise SystemExit("--mode jepegmix: not built yet")


def run_e2e(args, rig):
    raise SystemExit("--mode e2e: not built yet")
