"""margipose_amd -- the MargiPose forward/backward hot path on AMD MI355X (gfx950).

Public surface mirrors the reference for this path only:
    margipose_amd.models.MargiPoseModel / create_model / load_model
    margipose_amd.dsntnn.{flat_softmax, dsnt, js_reg_losses, euclidean_losses, average_loss, make_gauss}
"""
__version__ = '0.1.0'
