class AutoencoderKL:  # name only (type annotations in the reference)
    pass
