from .margipose_model import (CanonicalSkeletonDesc, Default_MargiPose_Desc, MargiPoseModel, create_model,  # noqa: F401
                              load_model)
from .chatterbox_model import ChatterboxModel, Default_Chatterbox_Desc  # noqa: F401
