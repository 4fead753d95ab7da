from .margipose_model import (CanonicalSkeletonDesc, Default_MargiPose_Desc, MargiPoseModel, create_model,  # noqa: F401
                              load_model)
