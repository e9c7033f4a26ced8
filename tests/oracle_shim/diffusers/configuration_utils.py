def register_to_config(fn):
    return fn
