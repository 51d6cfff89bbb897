from stemseg_amd.inference.online_chainer import OnlineChainer, TrackContainer, masks_to_coord_list  # noqa: F401
