from .gcn import GCN  # noqa: F401
