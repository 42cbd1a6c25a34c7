"""`python -m wsmgmap <script.py> [args...]` — run one of the reference's entry points (run.py) with the MI355X policy path in
place of `vlnce_baselines.models.policy` / `vlnce_baselines.common.aux_losses`, WITHOUT editing the reference
(wsmgmap.install(): common_trainer.py:24, dagger_trainer.py:25 resolve to this package)."""
import runpy
import sys

import wsmgmap


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m wsmgmap <script.py> [args...]")
    wsmgmap.install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))     # what `python script.py` does
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
