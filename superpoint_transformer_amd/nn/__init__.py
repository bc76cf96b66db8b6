"""Host-side mirror of the reference's ``src/nn`` operator surface: same class
names, constructor arguments, parameter names and forward signatures, with the
hot ops running on the HIP kernels of libspt_hip.so."""
from .attention import SelfAttentionBlock
from .dropout import DropPath
from .fusion import (AdditiveFusion, CatFusion, TakeFirstFusion, TakeSecondFusion,
                     fusion_factory)
from .mlp import FFN, MLP, Classifier
from .norm import INDEX_BASED_NORMS, GraphNorm, InstanceNorm, LayerNorm, UnitSphereNorm
from .pool import (AttentivePool, AttentivePoolWithLearntQueries, BaseAttentivePool, MaxPool,
                   MeanPool, MinPool, StdPool, SumPool, pool_factory)
from .stage import DownNFuseStage, PointStage, Stage, UpNFuseStage
from .transformer import TransformerBlock, VersionHolder
from .unpool import IndexUnpool
from .spt import SPT
