"""WanT2V pipeline surface (seaweed_apt/wan/text2video.py) — filled in after the DiT path."""


class WanT2V:
    def __init__(self, *a, **k):
        raise NotImplementedError("WanT2V: not built yet in this commit")
