"""CPU oracle for the COCO-DR contrastive dense-retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product path
(``coco-dr_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity status: the encoder arithmetic lives in the third-party ``transformers``
package (reference pins ``transformers==2.3.0``, warmup/commands/install.sh:2;
this container has 5.15.0).  The restatement here is pinned against golden
vectors produced in the build container by importing the reference's own
modules (COCO/modeling.py, ANCE/model/models.py) on top of the installed
``transformers`` BertModel in eager fp32 mode - see tests/golden/make_golden.py.
faiss / pytrec_eval are absent from the image: top-k and nDCG are "parity
unpinned" against those libraries (restated from their published definitions
and cross-checked on hand-computed cases; MRR@10 is pinned against the
reference's evaluate/evaluation/msmarco_eval.py).
"""
from .bert_oracle import *  # noqa: F401,F403
from .retrieval_oracle import *  # noqa: F401,F403
from .condenser_oracle import *  # noqa: F401,F403
from .optim_oracle import *  # noqa: F401,F403
from .idro_oracle import *  # noqa: F401,F403
from .collate_oracle import *  # noqa: F401,F403
