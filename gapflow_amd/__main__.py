"""Command line entry: ``python -m gapflow_amd -i input.yaml`` runs one problem to completion
(same flag as the reference's ``python -m GaPFlow -i``, GaPFlow/__main__.py:28-48)."""
import argparse
import sys

from . import Problem


def main(argv=None):
    cli = argparse.ArgumentParser(prog='python -m gapflow_amd',
                                  description="Advance a GaPFlow YAML problem on an MI355X.")
    cli.add_argument('-i', '--input', dest='filename', required=True, metavar='YAML', help="problem definition")
    cli.add_argument('--device', type=int, default=0, help="HIP device ordinal (default 0)")
    opts = cli.parse_args(argv)
    Problem.from_yaml(opts.filename).run()
    return 0


if __name__ == "__main__":
    sys.exit(main())
