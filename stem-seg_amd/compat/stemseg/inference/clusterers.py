from stemseg_amd.inference.clusterers import ClustererBase, SequentialClustering  # noqa: F401
