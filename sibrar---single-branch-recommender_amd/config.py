"""Configuration dataclasses of the SingleBranchNet plugin — same field names and defaults as the reference's
data/module_config_classes.py:19-25 (FeatureModuleConfig), :45-48 (SingleBranchFeatureConfig), :50-72
(EmbeddingRegularizationType), :76-95 (SingleBranchNetEntityConfig) and :98-127 (SingleBranchNetConfig), so that the YAML
``model:`` dictionaries of conf/single/algorithms/sbnet_*_conf.yml parse unchanged. (The reference parses them with
mashumaro; this is a dependency-free ``from_dict``.)"""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from enum import Enum
from typing import List, Optional, Set, Union


class EmbeddingRegularizationType(Enum):
    NoRegularization = 'no_regularization'
    PairwiseSingle = 'pairwise_single'
    CentralModality = 'central_modality'

    @classmethod
    def list(cls):
        return [c.value for c in cls]


class MissingField(KeyError):
    pass


def _from_dict(cls, d: dict):
    known = {f.name: f for f in fields(cls)}
    unknown = set(d) - set(known)
    if unknown:
        raise TypeError(f'{cls.__name__}: unknown configuration keys {sorted(unknown)}')
    for name, f in known.items():
        from dataclasses import MISSING
        if name not in d and f.default is MISSING and f.default_factory is MISSING:
            raise MissingField(f'{cls.__name__}: missing field "{name}"')
    return cls(**d)


@dataclass
class FeatureModuleConfig:
    feature_name: str
    embedding_dim: int
    pre_embedding_layers: Optional[List[int]] = None
    post_embedding_layers: Optional[List[int]] = None
    activation_fn: str = 'relu'

    @classmethod
    def from_dict(cls, d: dict):
        return _from_dict(cls, dict(d))

    def to_dict(self):
        return {f.name: getattr(self, f.name) for f in fields(self)}


@dataclass
class SingleBranchFeatureConfig:
    feature_name: str
    feature_hidden_layers: Optional[List[int]] = None

    @classmethod
    def from_dict(cls, d: dict):
        return _from_dict(cls, dict(d))


@dataclass
class SingleBranchNetEntityConfig:
    features: List[SingleBranchFeatureConfig]
    single_branch_hidden_layers: List[int]
    preference_hidden_layers: List[int]
    common_modality_dim: int
    activation_fn: str = 'relu'
    train_modalities: Optional[Set[str]] = None
    eval_modalities: Optional[Set[str]] = None
    sampling_seed: int = 42
    single_branch_input_dropout: Optional[float] = None
    aggregation_fn: str = 'mean'
    normalize_single_branch_input: bool = False
    embedding_regularization_type: EmbeddingRegularizationType = EmbeddingRegularizationType.NoRegularization
    central_modality: Optional[str] = None
    regularization_temperature: float = 1.
    regularization_weight: float = 1.
    apply_output_activation: bool = False
    apply_batch_normalization: bool = True
    apply_batch_norm_every: int = 0

    @classmethod
    def from_dict(cls, d: dict):
        d = dict(d)
        d['features'] = [f if isinstance(f, SingleBranchFeatureConfig) else SingleBranchFeatureConfig.from_dict(f)
                         for f in d.get('features', [])]
        for key in ('train_modalities', 'eval_modalities'):
            if d.get(key) is not None:
                d[key] = set(d[key])
        t = d.get('embedding_regularization_type')
        if t is not None and not isinstance(t, EmbeddingRegularizationType):
            d['embedding_regularization_type'] = EmbeddingRegularizationType(getattr(t, 'value', t))
        for key in ('single_branch_hidden_layers', 'preference_hidden_layers'):
            if d.get(key) is None and key in d:
                d[key] = []
        return _from_dict(cls, d)


@dataclass
class SingleBranchNetConfig:
    user: Union[SingleBranchNetEntityConfig, FeatureModuleConfig]
    item: Union[SingleBranchNetEntityConfig, FeatureModuleConfig]
    shared_common_dim: int

    @staticmethod
    def _conditional_parse_entity_conf(conf):
        """module_config_classes.py:114-119: try the plain feature-module config first, fall back to the entity config."""
        if not isinstance(conf, dict):
            return coerce_side_config(conf)
        try:
            return FeatureModuleConfig.from_dict(conf)
        except (MissingField, TypeError):
            return SingleBranchNetEntityConfig.from_dict(conf)

    @classmethod
    def from_dict(cls, d: dict):
        return cls(user=cls._conditional_parse_entity_conf(d['user']), item=cls._conditional_parse_entity_conf(d['item']),
                   shared_common_dim=d['shared_common_dim'])

    @property
    def is_user_sb_module(self) -> bool:
        return isinstance(self.user, SingleBranchNetEntityConfig)

    @property
    def is_item_sb_module(self) -> bool:
        return isinstance(self.item, SingleBranchNetEntityConfig)


def coerce_side_config(c):
    """Accept the reference's own config objects (duck-typed) as well as ours."""
    if isinstance(c, (SingleBranchNetEntityConfig, FeatureModuleConfig)):
        return c
    if hasattr(c, 'features'):
        d = {f.name: getattr(c, f.name) for f in fields(SingleBranchNetEntityConfig) if hasattr(c, f.name)}
        d['features'] = [{'feature_name': f.feature_name, 'feature_hidden_layers': f.feature_hidden_layers} for f in c.features]
        return SingleBranchNetEntityConfig.from_dict(d)
    return FeatureModuleConfig(c.feature_name, c.embedding_dim, getattr(c, 'pre_embedding_layers', None),
                               getattr(c, 'post_embedding_layers', None), getattr(c, 'activation_fn', 'relu'))


def coerce_net_config(c) -> SingleBranchNetConfig:
    if isinstance(c, SingleBranchNetConfig):
        return c
    if isinstance(c, dict):
        return SingleBranchNetConfig.from_dict(c)
    return SingleBranchNetConfig(coerce_side_config(c.user), coerce_side_config(c.item), c.shared_common_dim)


# ---- DropoutNet (data/module_config_classes.py:10-42) ---------------------------------------------------------------------------
class DropoutNetSamplingStrategy(Enum):
    Normal = 1            # the reference uses enum.auto(): 1, 2
    NoPreference = 2

    @classmethod
    def list(cls):
        return [c.value for c in cls]


@dataclass
class DropoutNetEntityConfig:
    features: List[FeatureModuleConfig]
    preference_layers: List[int]          # number of items / users is prepended automatically
    common_hidden_layers: List[int]       # content + preference dim in front, shared common dim behind
    activation_fn: str = 'relu'

    @classmethod
    def from_dict(cls, d: dict):
        d = dict(d)
        d['features'] = [f if isinstance(f, FeatureModuleConfig) else FeatureModuleConfig.from_dict(f) for f in d.get('features', [])]
        return _from_dict(cls, d)

    def to_dict(self):
        return {'features': [f.to_dict() for f in self.features], 'preference_layers': self.preference_layers,
                'common_hidden_layers': self.common_hidden_layers, 'activation_fn': self.activation_fn}


@dataclass
class DropoutNetConfig:
    user: DropoutNetEntityConfig
    item: DropoutNetEntityConfig
    shared_common_dim: int
    sampling_seed: int = 42

    @classmethod
    def from_dict(cls, d: dict):
        d = dict(d)
        for side in ('user', 'item'):
            if side in d and not isinstance(d[side], DropoutNetEntityConfig):
                d[side] = DropoutNetEntityConfig.from_dict(d[side])
        return _from_dict(cls, d)

    def to_dict(self):
        return {'user': self.user.to_dict(), 'item': self.item.to_dict(), 'shared_common_dim': self.shared_common_dim,
                'sampling_seed': self.sampling_seed}
