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
from .io import GraphDataset, PrefetchLoader, ResidentDataset, load_graph, renumber_nodes
from .edge_classifier import ECForGraphTCN, PerfectEdgeClassification
from .interaction_network import InteractionNetwork
from .graph_construction import MLGraphConstruction, MLPCTransformer, knn_scan, knn_with_max_radius
from .graph_masks import get_good_node_mask, get_good_node_mask_tensors
from .losses_ec import (EdgeWeightBCELoss, EdgeWeightFocalLoss, HaughtyFocalLoss, binary_focal_loss,
                        falsify_low_pt_edges)
from .losses_ml import GraphConstructionHingeEmbeddingLoss
from .losses_oc import CondensationLossRG, CondensationLossTiger, MultiLossFctReturn
from .mlp import MLP
from .locality import node_order
from .precision import bf16_storage
from .resin import ResIN
from .postprocessing import DBSCANFastRescan, dbscan
from .track_condensation_networks import (GraphConstructionFCNN, GraphConstructionHeteroEncResFCNN,
                                            GraphConstructionHeteroResFCNN, GraphConstructionResIN, GraphTCN,
                                            GraphTCNForMLGCPipeline, PerfectECGraphTCN,
                                            HeterogeneousResFCNN, ModularGraphTCN, PreTrainedECGraphTCN,
                                            ResFCNN)

__version__ = "0.2.0"
__all__ = ["Data", "collate", "MLP", "InteractionNetwork", "ResIN", "ECForGraphTCN",
           "EdgeWeightBCELoss", "falsify_low_pt_edges", "MLGraphConstruction",
           "knn_with_max_radius", "get_good_node_mask", "get_good_node_mask_tensors",
           "CondensationLossRG", "CondensationLossTiger", "MultiLossFctReturn", "bf16_storage", "node_order", "GraphTCN", "ModularGraphTCN",
           "PreTrainedECGraphTCN", "ResFCNN", "GraphConstructionHingeEmbeddingLoss",
           "GraphConstructionFCNN", "HeterogeneousResFCNN", "GraphConstructionHeteroResFCNN",
           "GraphConstructionHeteroEncResFCNN", "GraphConstructionResIN", "PerfectECGraphTCN",
           "GraphTCNForMLGCPipeline", "PerfectEdgeClassification", "MLPCTransformer", "knn_scan", "EdgeWeightFocalLoss", "HaughtyFocalLoss", "binary_focal_loss", "DBSCANFastRescan", "dbscan", "load_graph", "GraphDataset", "PrefetchLoader", "ResidentDataset", "renumber_nodes"]
