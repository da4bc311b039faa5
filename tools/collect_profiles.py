"""Copy the merged gpurun_out/round/* artefacts of tools/profile_round.sh into profiles/ under a round prefix."""
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(repo, "gpurun_out", "round"), os.path.join(repo, "profiles")
for f in sorted(os.listdir(src)):
    if f.endswith((".json", ".csv", ".txt")) and os.path.getsize(os.path.join(src, f)) > 0:
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
        print(f"{tag}_{f}")
