#!/usr/bin/env python
"""Export the Discogs label vocabularies (DATA: two lists of strings,
/root/reference/models/discogs_labels.py:1,404) to plain-text tables, one label per line,
consumed by ``maest_amd.labels``.  Run in the authoring container only."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/models")
import discogs_labels as L  # noqa: E402

out = os.path.join(REPO, "maest_amd", "data")
os.makedirs(out, exist_ok=True)
for name, lst in (("discogs_400labels", L.discogs_400labels), ("discogs_519labels", L.discogs_519labels)):
    assert all("\n" not in s for s in lst)
    with open(os.path.join(out, name + ".txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lst) + "\n")
    print(name, len(lst))
