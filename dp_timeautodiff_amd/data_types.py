"""Column typing of the reference's tabular parser — only what `train_dp_autoencoder` needs from it.

The reference's DP loop opens with `parser = pce.DataFrameParser().fit(real_df, threshold)` and uses nothing of the
parser but `datatype_info()` = {n_bins, n_cats, n_nums, cards} (Model_Code/timeautoencoder.py:626-634): the sizes of
the VAE's embedding tables and output heads.  The parser itself (encoders, scaling, inverse transforms:
Model_Code/process_edited.py) is host preprocessing and out of scope; this module restates its
column-typing rule (process_edited.py:107-160) so that the harness can take the same `real_df` argument.
Checked against the reference parser on its own CSVs in tests/test_reference_dropin.py (build container only).
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np


def datatype_info(real_df, threshold: float) -> Dict[str, Any]:
    """{n_bins, n_cats, n_nums, cards} of a pandas DataFrame, by the reference's rule:
       object/str column: <= 2 distinct values (NaN counted) -> binary, else categorical;
       integer or float64 column: <= 2 distinct -> binary, 3..25 distinct -> categorical, else numerical — and a
       numerical column in which k values each fill more than `threshold` of the rows also spawns a derived column:
       binary for k = 1, categorical (k + 1 classes) for 2 <= k <= 25."""
    n_bins, n_nums, cards = 0, 0, []
    derived_cards = []
    for column, dtype in real_df.dtypes.to_dict().items():
        col = real_df[column]
        if dtype in ("O", "<U32"):
            if col.nunique(dropna=False) <= 2:
                n_bins += 1
            else:
                cards.append(int(col.astype(str).nunique()))
            continue
        is_int, is_f64 = np.issubdtype(dtype, np.integer), np.issubdtype(dtype, np.float64)
        nun = col.nunique()
        if (is_int or is_f64) and nun <= 2:
            n_bins += 1
        elif (is_int or is_f64) and 3 <= nun <= 25:
            cards.append(int(col.astype(int).nunique()))
        else:
            n_nums += 1
            counts = col.value_counts()
            k = int((counts > threshold * len(col)).sum())
            if k == 1:
                n_bins += 1
            elif 2 <= k <= 25:
                repeated = counts[counts > threshold * len(col)].index.tolist()
                derived = col.apply(lambda x: repeated.index(x) + 1 if x in repeated else 0)
                derived_cards.append((column, int(derived.nunique())))
    # the parser appends derived categorical columns in column order as it meets them, i.e. interleaved with the plain
    # ones; both kinds are appended to the same list in one pass, so a single pass in column order reproduces `cards`
    if derived_cards:
        cards = _cards_in_column_order(real_df, threshold)
    return {"n_bins": n_bins, "n_cats": len(cards), "n_nums": n_nums, "cards": cards}


def _cards_in_column_order(real_df, threshold):
    cards = []
    for column, dtype in real_df.dtypes.to_dict().items():
        col = real_df[column]
        if dtype in ("O", "<U32"):
            if col.nunique(dropna=False) > 2:
                cards.append(int(col.astype(str).nunique()))
            continue
        is_int, is_f64 = np.issubdtype(dtype, np.integer), np.issubdtype(dtype, np.float64)
        nun = col.nunique()
        if (is_int or is_f64) and nun <= 2:
            continue
        if (is_int or is_f64) and 3 <= nun <= 25:
            cards.append(int(col.astype(int).nunique()))
            continue
        counts = col.value_counts()
        repeated = counts[counts > threshold * len(col)].index.tolist()
        if 2 <= len(repeated) <= 25:
            derived = col.apply(lambda x: repeated.index(x) + 1 if x in repeated else 0)
            cards.append(int(derived.nunique()))
    return cards


def resolve(real_df, threshold) -> Dict[str, Any]:
    """What `train_dp_autoencoder` accepts in the reference's `real_df` position: the DataFrame itself, a fitted
    parser (anything with `.datatype_info()`), or the {n_bins, n_cats, n_nums, cards} dict directly."""
    if isinstance(real_df, dict):
        info = real_df
    elif hasattr(real_df, "datatype_info"):
        info = real_df.datatype_info()
    elif hasattr(real_df, "dtypes"):
        info = datatype_info(real_df, threshold)
    else:
        raise TypeError("real_df must be a pandas DataFrame, a fitted parser with .datatype_info(), or a dict with "
                        "n_bins / n_cats / n_nums / cards")
    missing = {"n_bins", "n_cats", "n_nums", "cards"} - set(info)
    if missing:
        raise ValueError(f"datatype info lacks {sorted(missing)}")
    return {"n_bins": int(info["n_bins"]), "n_cats": int(info["n_cats"]), "n_nums": int(info["n_nums"]),
            "cards": [int(c) for c in info["cards"]]}
