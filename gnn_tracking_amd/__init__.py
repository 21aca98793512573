"""gnn_tracking_amd - MI355X (gfx950) native hot path of gnn_tracking.

Interaction-network message passing, the ``ECForGraphTCN`` edge classifier, kNN graph
construction and the object-condensation loss reductions, behind the reference's own
``torch.nn.Module`` / PyG-``Data`` operator surface, executed by hand-written HIP
kernels through the C ABI of ``libgnntrk.so`` (``include/gnntrk.h``).

Importing the package does not touch the GPU; the first operator call loads (and, if
needed, builds) the HIP extension and fails loudly when that is impossible - there is
no CPU fallback.
"""

from .data import Data, collate
from .edge_classifier import ECForGraphTCN
from .interaction_network import InteractionNetwork
from .losses_ec import EdgeWeightBCELoss, falsify_low_pt_edges
from .mlp import MLP
from .resin import ResIN

__version__ = "0.1.0"
__all__ = ["Data", "collate", "MLP", "InteractionNetwork", "ResIN", "ECForGraphTCN",
           "EdgeWeightBCELoss", "falsify_low_pt_edges"]
