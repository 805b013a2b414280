"""``python -m gapflow_amd -i input.yaml`` (GaPFlow/__main__.py:28-48)."""
from argparse import ArgumentParser

from . import Problem


def get_parser():
    parser = ArgumentParser()
    required = parser.add_argument_group('required arguments')
    required.add_argument('-i', '--input', dest="filename", help="YAML input file", required=True)
    return parser


if __name__ == "__main__":
    args = get_parser().parse_args()
    Problem.from_yaml(args.filename).run()
