class InputFeature:  # VAE-encoder-only placeholder
    def __init__(self, *a, **k):
        pass
