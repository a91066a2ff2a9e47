"""tasks/run.py -- the reference's launch target (examples/openqa/emdr2_*.sh end in `... tasks/run.py ${OPTIONS} ${CONFIG_ARGS}`,
reference tasks/run.py:49-67).  Same flags, one process per GPU; everything is emdr2_amd.tasks.run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from emdr2_amd.tasks.run import initialize, main  # noqa: E402,F401

if __name__ == '__main__':
    main(sys.argv[1:])
