"""Generates tests/golden/script_flags.json: the command-line flag lists the reference's launch scripts pass to tasks/run.py
(examples/openqa/emdr2_{nq,trivia,webq}.sh: the OPTIONS block + config_base's CONFIG_ARGS), with the shell variables expanded the way bash
would.  Data only -- flag lists, no script text.  Run in the build container:  python tests/golden/gen_script_flags.py"""
import json
import os
import re
import string

REF = "/root/reference/examples/openqa"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "script_flags.json")


def expand(script):
    text = open(os.path.join(REF, script)).read()
    env = {}
    for m in re.finditer(r'^([A-Z_]+)=("?)([^\n"]*)\2\s*$', text, re.M):            # simple NAME="value" lines, in order
        env[m.group(1)] = string.Template(m.group(3)).safe_substitute(env)
    config = re.search(r'export CONFIG_ARGS="(.*?)"', text, re.S).group(1)
    options = re.search(r'^OPTIONS="(.*?)"\s*$', text, re.S | re.M).group(1)
    flat = lambda s: " ".join(string.Template(s.replace("\\\n", " ")).safe_substitute(env).split())
    launcher = re.search(r'DISTRIBUTED_ARGS="(.*?)"', text).group(1)
    return {"argv": (flat(options) + " " + flat(config)).split(), "nproc_per_node": int(re.search(r"--nproc_per_node (\d+)", launcher).group(1))}


if __name__ == "__main__":
    out = {s: expand(s) for s in ("emdr2_nq.sh", "emdr2_trivia.sh", "emdr2_webq.sh")}
    json.dump(out, open(OUT, "w"), indent=1)
    for k, v in out.items():
        print(k, len(v["argv"]), "tokens")
