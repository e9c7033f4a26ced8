class BaseOutput:
    pass
