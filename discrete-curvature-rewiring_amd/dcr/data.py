"""Minimal duck-typed stand-in for ``torch_geometric.data.Data``.

The reference passes PyG ``Data`` objects across its call surface
(rewiring/rewire.py:7, models/gcn.py:32, experiment/training_loop.py:51,71).
PyG is not part of this image, so the rewiring and GCN entry points accept any
object with these attributes; this class is the one the build's own drivers
and tests use.  Attribute access and ``data['val_mask']`` item access both
work, as training_loop.py:71 requires.
"""


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, num_nodes=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self.y = y
        self._num_nodes = num_nodes
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        if self._num_nodes is not None:
            return int(self._num_nodes)
        if self.x is not None:
            return int(self.x.shape[0])
        if self.edge_index is not None and self.edge_index.shape[1] > 0:
            return int(self.edge_index.max()) + 1
        return 0

    @num_nodes.setter
    def num_nodes(self, v):
        self._num_nodes = v

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return hasattr(self, key) and getattr(self, key) is not None

    def keys(self):
        return [k for k, v in self.__dict__.items() if not k.startswith('_') and v is not None]

    def to(self, device):
        for k in list(self.__dict__):
            v = self.__dict__[k]
            if hasattr(v, 'to') and hasattr(v, 'device'):
                self.__dict__[k] = v.to(device)
        return self

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if hasattr(v, 'shape') else f"{k}={v!r}")
        return "Data(" + ", ".join(parts) + ")"


class Dataset:
    """What models/gcn.py:16 reads from its ``dataset`` argument."""

    def __init__(self, data, num_classes):
        self.data = data
        self.num_classes = int(num_classes)
