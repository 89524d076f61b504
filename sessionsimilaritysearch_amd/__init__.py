"""MI355X-native session-similarity retrieval hot path.

Drop-in surface of the reference's path:
  normalize, build_index, FlatIndex           (index.py   <- test_amazon_filterd.py / util_amazon_filtered.py)
  SessionEncoder                              (encoder.py <- model/model.py UnifyPoolingGraphLevelEncoder)
  SessionBatch, build_batch, synthetic_actions (sessions.py <- sequence_to_graph + Batch.from_data_list)
"""
from ._lib import SssError, build, exported_symbols, lib  # noqa: F401

__all__ = ["SssError", "build", "exported_symbols", "lib"]
