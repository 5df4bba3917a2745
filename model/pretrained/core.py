from t2ms_amd.model.pretrained.core import BaseModel  # noqa: F401
