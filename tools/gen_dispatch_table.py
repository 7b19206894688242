"""Writes docs/DISPATCH.md: the rollout kernel instance per registry layout and launch shape, as oc_rollout_plan answers
(overcooked_ai_amd/dispatch.py).  Needs the built library, no GPU.  tests/test_dispatch_table.py fails when the file is stale."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from overcooked_ai_amd import dispatch  # noqa: E402

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "docs", "DISPATCH.md")
    text = dispatch.render(dispatch.table())
    with open(path, "w") as f:
        f.write(text)
    print("wrote %s (%d rows)" % (os.path.normpath(path), text.count("\n") - 7))
