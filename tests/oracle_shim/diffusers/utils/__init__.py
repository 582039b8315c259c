import logging as _logging


class BaseOutput:
    pass


class logging:  # noqa: N801 - mimics ``diffusers.utils.logging``
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)
