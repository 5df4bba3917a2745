from t2ms_amd.model.backbone.rectified_flow import RectifiedFlow  # noqa: F401
