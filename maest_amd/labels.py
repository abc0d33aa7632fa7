"""Discogs label vocabularies returned verbatim by ``MAEST.predict_labels``
(reference: models/discogs_labels.py:1 and :404, consumed at models/maest.py:501-504,939).
Stored as plain-text tables (one label per line) under maest_amd/data/."""
import os

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _load(name):
    with open(os.path.join(_DIR, name + ".txt"), encoding="utf-8") as f:
        return f.read().split("\n")[:-1]


discogs_400labels = _load("discogs_400labels")
discogs_519labels = _load("discogs_519labels")
assert len(discogs_400labels) == 400 and len(discogs_519labels) == 519
