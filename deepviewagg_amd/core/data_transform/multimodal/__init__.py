from .image import MapImages  # noqa: F401
