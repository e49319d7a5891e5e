"""A deliberately tiny column store for node/edge attribute tables."""
from .series import Series
from .dataframe import DataFrame

__all__ = ['Series', 'DataFrame']
