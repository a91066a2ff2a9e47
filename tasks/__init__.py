"""`tasks` -- the reference's task entry points over emdr2_amd (`python tasks/run.py --task OPENQA ...`, tasks/run.py:49-67)."""
